// Flash-style fused attention for gfx950 (forward, dQ backward, dK/dV backward) on 32x32x16 bf16 MFMA.
//
// Replaces: eva_vit.py:128-145 (ViT, head_dim 88, scale 88^-0.5, no mask), Qformer.py:195-262 (self 32x32 and
// cross 32x257, scale 1/8, dropout on probabilities), modeling_t5.py:536-603 (T5: NO 1/sqrt(d) scaling, shared
// relative-position bias computed on the fly from a 257-entry per-head LUT instead of the reference's dense
// [1,H,S,S] tensor, key-padding / causal masks, fp32 softmax, dropout on probabilities).
//
// Design: one wave owns 32 query rows (dKV kernel: 32 key rows) and never touches LDS for operands: the
// "swapped" product S^T = K Q^T makes every lane own ONE query column, so the softmax reductions are in-lane
// plus one lane^32 exchange, and the fp32 score registers, after bf16 packing, ARE the B operand of the second
// product O^T = V^T P^T — provided the 32 keys of a tile are fed to the first MFMA in the row order pi(i)
// (bits 2 and 3 of the row index swapped), which is a free address permutation.  V^T (and K^T, Q^T, dO^T for
// the backward) are [B,H,DP,Spad] transposed copies produced by mrblip_head_transpose, zero padded so padded
// keys/dims contribute exact zeros.  K/V tiles are re-read by every wave from L2 (per-head K+V are L2-resident).
// The per-element work is specialised at compile time (FLAGS: LUT bias / key mask / causal / dropout / key-split)
// and is branch-free: loads use clamped addresses + selects, the 16 LUT reads of a tile are issued together, the
// key mask arrives as four 16-B loads per tile.  With few queries (decoder cross-attention, Sq <= 32) the four
// waves of a block split the key range and merge their partial (m, l, O) / dQ through LDS.
#include "common.h"

struct T4 {  // element (b,h,s,d) at ptr + b*bs + h*hs + s*rs + d   (row-major in d)
  const bf16_t* ptr;
  long long bs, hs, rs;
};
struct T4T {  // element (b,h,d,s) at ptr + b*bs + h*hs + d*ds + s   (transposed copy)
  const bf16_t* ptr;
  long long bs, hs, ds;
};

struct AttnArgs {
  T4 Q, K, V, O, dO, dQ, dK, dV;
  T4T Vt, Kt, Qt, dOt;
  float* LSE;          // [B,H,Sqpad]
  float* Delta;        // [B,H,Sqpad]
  const float* lut;    // [H,257] relative-position bias by clamp(key - q, -128, 128) + 128, or nullptr
  const int* kmask;    // [B,Skpad] 1 = attend (padded to a multiple of 32 ints per row), or nullptr
  int B, H, Sq, Sk, D, Sqpad, Skpad;
  float scale;
  DropoutArg drop;
};

enum { F_LUT = 1, F_MASK = 2, F_CAUSAL = 4, F_DROP = 8, F_SPLIT = 16 };

// attention-probability dropout draws: one 32-bit hash per (row, key pair): index = row * ceil(Sk/2) + key/2, the low 16 bits
// serve the even key and the high 16 bits the odd key -> 8 hashes per 16 scores where a lane owns consecutive keys.
__device__ __forceinline__ uint32_t attn_drop_hash(uint32_t row, int key, int skh, uint32_t seed, uint32_t site) {
  return mrb_hash(row * (uint32_t)skh + (uint32_t)(key >> 1), seed, site);
}
#define NEG_BIG (-1.0e30f)

__device__ __forceinline__ int perm23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 sel8(bool ok, bf16x8 v) {
  const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  return ok ? v : z;
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  union { bf16x8 v8; uint32_t u[4]; } r;
  r.u[0] = pack2bf(v[0], v[1]); r.u[1] = pack2bf(v[2], v[3]); r.u[2] = pack2bf(v[4], v[5]); r.u[3] = pack2bf(v[6], v[7]);
  return r.v8;
}
__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// rows of a row-major [S, D] head slice as MFMA fragments: lane (row, hi) gets d = 16 s + 8 hi .. +7, zero outside.
// Branch-free: the address is clamped into the tensor and the value selected afterwards.
template <int KS>
__device__ __forceinline__ void load_rows(bf16x8 (&f)[KS], const bf16_t* base, long long rs, int row, int nrows, int D, int hi) {
  const bool r_ok = row < nrows;
  const bf16_t* rp = base + (long long)min(row, nrows - 1) * rs;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int d0 = 16 * s + 8 * hi;
    f[s] = sel8(r_ok && d0 < D, ld8(rp + min(d0, D - 8)));
  }
}

// valid-key bits of one 32-key tile for this lane's 16 keys (k0 + 16c + 8hi + j), from four 16-B loads
__device__ __forceinline__ uint32_t mask_bits(const int* km, int k0, int hi) {
  uint32_t bits = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int4 a = *reinterpret_cast<const int4*>(km + k0 + 16 * c + 8 * hi), b = *reinterpret_cast<const int4*>(km + k0 + 16 * c + 8 * hi + 4);
    bits |= (uint32_t)(a.x != 0) << (8 * c + 0); bits |= (uint32_t)(a.y != 0) << (8 * c + 1);
    bits |= (uint32_t)(a.z != 0) << (8 * c + 2); bits |= (uint32_t)(a.w != 0) << (8 * c + 3);
    bits |= (uint32_t)(b.x != 0) << (8 * c + 4); bits |= (uint32_t)(b.y != 0) << (8 * c + 5);
    bits |= (uint32_t)(b.z != 0) << (8 * c + 6); bits |= (uint32_t)(b.w != 0) << (8 * c + 7);
  }
  return bits;
}

template <int DP, int FLAGS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP, SPLIT = FLAGS & F_SPLIT;
  __shared__ float lut[LUT ? 257 : 1];
  __shared__ float red[SPLIT ? 3 * (MT * 16 + 2) * 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = p.lut[h * 257 + i];
    __syncthreads();
  }
  const int q0 = SPLIT ? blockIdx.x * 32 : (blockIdx.x * 4 + w) * 32;
  if (!SPLIT && q0 >= p.Sq) return;
  const int q = q0 + l31;
  const bool q_ok = q < p.Sq;
  bf16x8 qf[KS];
  load_rows<KS>(qf, p.Q.ptr + b * p.Q.bs + h * p.Q.hs, p.Q.rs, q, p.Sq, p.D, hi);
  float m_run = NEG_BIG, l_run = 0.f;
  f32x16 o[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(o[mt]);
  const int kend = CAUSAL ? min(p.Sk, q0 + 32) : p.Sk;
  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vtbase = p.Vt.ptr + b * p.Vt.bs + h * p.Vt.hs;
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const int skh = (p.Sk + 1) >> 1;
  const int kstart = SPLIT ? w * 32 : 0, kstep = SPLIT ? 128 : 32;

  // software pipeline: K fragments of the next tile and V^T fragments of this tile are in flight during the score math
  bf16x8 kcur[KS];
  load_rows<KS>(kcur, kbase, p.K.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  for (int k0 = kstart; k0 < kend; k0 += kstep) {
    bf16x8 vf[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16_t* vp = vtbase + (long long)(mt * 32 + l31) * p.Vt.ds + k0 + 8 * hi;
      vf[mt][0] = ld8(vp);
      vf[mt][1] = ld8(vp + 16);
    }
    uint32_t vmask = 0xffffu;
    if (MASK) vmask = mask_bits(km, k0, hi);
    bf16x8 knext[KS];
    load_rows<KS>(knext, kbase, p.K.rs, k0 + kstep + perm23(l31), p.Sk, p.D, hi);
    f32x16 sacc;
    zero16(sacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur[s], qf[s], sacc, 0, 0, 0);
    // lane (q, hi), register r  <->  key k0 + 16*(r>>3) + 8*hi + (r&7)
    float sv[16];
    if (LUT) {
      // every |key - q| >= 128 shares one bucket: tiles entirely beyond that distance need one LUT value, not sixteen reads
      if (k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128) {
        const float bconst = lut[k0 > q0 ? 256 : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bconst;
      } else {
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rel = (k0 + 16 * (r >> 3) + 8 * hi + (r & 7)) - q;
          bias[r] = lut[max(-128, min(128, rel)) + 128];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bias[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale;
    }
    float mx = NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
      bool ok = key < p.Sk;
      if (CAUSAL) ok = ok && (key <= q);
      if (!ok) vmask &= ~(1u << r);
      mx = fmaxf(mx, ((vmask >> r) & 1u) ? sv[r] : NEG_BIG);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = ((vmask >> r) & 1u) ? __expf(sv[r] - m_new) : 0.f;
      psum += e;
      pv[r] = e;
    }
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t hsh = attn_drop_hash(row_id, k0 + 16 * (r >> 3) + 8 * hi + (r & 7), skh, drop_seed, p.drop.site);
        pv[r] = (hsh & 0xffffu) >= p.drop.thresh24 ? pv[r] * p.drop.inv_keep : 0.f;
        pv[r + 1] = (hsh >> 16) >= p.drop.thresh24 ? pv[r + 1] * p.drop.inv_keep : 0.f;
      }
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    const bf16x8 pf0 = pack8(pv), pf1 = pack8(pv + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[mt][0], pf0, o[mt], 0, 0, 0);
      o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[mt][1], pf1, o[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) kcur[s] = knext[s];
  }
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (SPLIT) {  // merge the four key-range partials: wave 0 collects (m, l, O) of waves 1..3
    constexpr int STR = (MT * 16 + 2) * 64;
    if (w > 0) {
      float* r = red + (w - 1) * STR;
      r[lane] = m_run;
      r[64 + lane] = l_tot;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) r[(2 + mt * 16 + i) * 64 + lane] = o[mt][i];
    }
    __syncthreads();
    if (w > 0) return;
    float m_all = m_run;
#pragma unroll
    for (int j = 0; j < 3; ++j) m_all = fmaxf(m_all, red[j * STR + lane]);
    const float f0 = __expf(m_run - m_all);
    l_tot *= f0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[mt][i] *= f0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* r = red + j * STR;
      const float fj = __expf(r[lane] - m_all);
      l_tot += r[64 + lane] * fj;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[mt][i] += r[(2 + mt * 16 + i) * 64 + lane] * fj;
    }
    m_run = m_all;
  }
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_ok) {
    bf16_t* op = const_cast<bf16_t*>(p.O.ptr) + b * p.O.bs + h * p.O.hs + (long long)q * p.O.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D)
          *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv),
                                                          pack2bf(o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv));
      }
    if (p.LSE && hi == 0) p.LSE[((long long)(b * p.H + h)) * p.Sqpad + q] = m_run + __logf(fmaxf(l_tot, 1e-37f));
  }
}

// ---- backward, part 1: dQ (and Delta = rowsum(dO * O), needed by part 2).  Same ownership as the forward.
template <int DP, int FLAGS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP, SPLIT = FLAGS & F_SPLIT;
  __shared__ float lut[LUT ? 257 : 1];
  __shared__ float red[SPLIT ? 3 * MT * 16 * 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = p.lut[h * 257 + i];
    __syncthreads();
  }
  const int q0 = SPLIT ? blockIdx.x * 32 : (blockIdx.x * 4 + w) * 32;
  if (!SPLIT && q0 >= p.Sq) return;
  const int q = q0 + l31;
  const bool q_ok = q < p.Sq;
  bf16x8 qf[KS], dof[KS];
  float delta = 0.f;
  {
    load_rows<KS>(qf, p.Q.ptr + b * p.Q.bs + h * p.Q.hs, p.Q.rs, q, p.Sq, p.D, hi);
    load_rows<KS>(dof, p.dO.ptr + b * p.dO.bs + h * p.dO.hs, p.dO.rs, q, p.Sq, p.D, hi);
    bf16x8 of[KS];
    load_rows<KS>(of, p.O.ptr + b * p.O.bs + h * p.O.hs, p.O.rs, q, p.Sq, p.D, hi);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) delta += bf2f((bf16_t)dof[s][j]) * bf2f((bf16_t)of[s][j]);
  }
  delta += __shfl_xor(delta, 32, 64);
  const long long stat_off = ((long long)(b * p.H + h)) * p.Sqpad + min(q, p.Sqpad - 1);
  const float lse = q_ok ? p.LSE[stat_off] : 0.f;
  if (q_ok && hi == 0 && (!SPLIT || w == 0)) p.Delta[stat_off] = delta;

  f32x16 dq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(dq[mt]);
  const int kend = CAUSAL ? min(p.Sk, q0 + 32) : p.Sk;
  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vbase = p.V.ptr + b * p.V.bs + h * p.V.hs;
  const bf16_t* ktbase = p.Kt.ptr + b * p.Kt.bs + h * p.Kt.hs;
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const int skh = (p.Sk + 1) >> 1;
  const int kstart = SPLIT ? w * 32 : 0, kstep = SPLIT ? 128 : 32;

  bf16x8 kcur[KS], vcur[KS];
  load_rows<KS>(kcur, kbase, p.K.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  load_rows<KS>(vcur, vbase, p.V.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  for (int k0 = kstart; k0 < kend; k0 += kstep) {
    bf16x8 ktf[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16_t* kt = ktbase + (long long)(mt * 32 + l31) * p.Kt.ds + k0 + 8 * hi;
      ktf[mt][0] = ld8(kt);
      ktf[mt][1] = ld8(kt + 16);
    }
    uint32_t vmask = 0xffffu;
    if (MASK) vmask = mask_bits(km, k0, hi);
    bf16x8 knext[KS], vnext[KS];
    load_rows<KS>(knext, kbase, p.K.rs, k0 + kstep + perm23(l31), p.Sk, p.D, hi);
    load_rows<KS>(vnext, vbase, p.V.rs, k0 + kstep + perm23(l31), p.Sk, p.D, hi);
    f32x16 sacc, dpacc;
    zero16(sacc);
    zero16(dpacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kcur[s], qf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vcur[s], dof[s], dpacc, 0, 0, 0);
    }
    float sv[16];
    if (LUT) {
      // every |key - q| >= 128 shares one bucket: tiles entirely beyond that distance need one LUT value, not sixteen reads
      if (k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128) {
        const float bconst = lut[k0 > q0 ? 256 : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bconst;
      } else {
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rel = (k0 + 16 * (r >> 3) + 8 * hi + (r & 7)) - q;
          bias[r] = lut[max(-128, min(128, rel)) + 128];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bias[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale;
    }
    float ds[16], dpv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dpv[r] = dpacc[r];
    if (DROP) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t hsh = attn_drop_hash(row_id, k0 + 16 * (r >> 3) + 8 * hi + (r & 7), skh, drop_seed, p.drop.site);
        dpv[r] = (hsh & 0xffffu) >= p.drop.thresh24 ? dpv[r] * p.drop.inv_keep : 0.f;
        dpv[r + 1] = (hsh >> 16) >= p.drop.thresh24 ? dpv[r + 1] * p.drop.inv_keep : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
      bool ok = q_ok && key < p.Sk && ((vmask >> r) & 1u);
      if (CAUSAL) ok = ok && (key <= q);
      const float pr = ok ? __expf(sv[r] - lse) : 0.f;
      ds[r] = ok ? pr * (dpv[r] - delta) * p.scale : 0.f;
    }
    const bf16x8 f0 = pack8(ds), f1 = pack8(ds + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[mt][0], f0, dq[mt], 0, 0, 0);
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[mt][1], f1, dq[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { kcur[s] = knext[s]; vcur[s] = vnext[s]; }
  }
  if (SPLIT) {
    if (w > 0) {
      float* r = red + (w - 1) * MT * 16 * 64;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) r[(mt * 16 + i) * 64 + lane] = dq[mt][i];
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[mt][i] += red[(j * MT * 16 + mt * 16 + i) * 64 + lane];
  }
  if (q_ok) {
    bf16_t* op = const_cast<bf16_t*>(p.dQ.ptr) + b * p.dQ.bs + h * p.dQ.hs + (long long)q * p.dQ.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D)
          *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(dq[mt][4 * g], dq[mt][4 * g + 1]), pack2bf(dq[mt][4 * g + 2], dq[mt][4 * g + 3]));
      }
  }
}

// ---- backward, part 2: dK, dV.  One wave owns 32 keys and walks the query tiles.
template <int DP, int FLAGS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP;
  __shared__ float lut[LUT ? 257 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = p.lut[h * 257 + i];
    __syncthreads();
  }
  const int kb0 = (blockIdx.x * 4 + w) * 32;
  if (kb0 >= p.Sk) return;
  const int key = kb0 + l31;
  bool key_ok = key < p.Sk;
  bf16x8 kf[KS], vf[KS];
  load_rows<KS>(kf, p.K.ptr + b * p.K.bs + h * p.K.hs, p.K.rs, key, p.Sk, p.D, hi);
  load_rows<KS>(vf, p.V.ptr + b * p.V.bs + h * p.V.hs, p.V.rs, key, p.Sk, p.D, hi);
  if (MASK) key_ok = key_ok && (p.kmask[(long long)b * p.Skpad + key] != 0);
  f32x16 dk[MT], dv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { zero16(dk[mt]); zero16(dv[mt]); }
  const bf16_t* qbase = p.Q.ptr + b * p.Q.bs + h * p.Q.hs;
  const bf16_t* dobase = p.dO.ptr + b * p.dO.bs + h * p.dO.hs;
  const bf16_t* qtbase = p.Qt.ptr + b * p.Qt.bs + h * p.Qt.hs;
  const bf16_t* dotbase = p.dOt.ptr + b * p.dOt.bs + h * p.dOt.hs;
  const float* lsebase = p.LSE + ((long long)(b * p.H + h)) * p.Sqpad;
  const float* delbase = p.Delta + ((long long)(b * p.H + h)) * p.Sqpad;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t bh_idx = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq;
  const int qstart = CAUSAL ? kb0 : 0;

  bf16x8 qcur[KS], docur[KS];
  load_rows<KS>(qcur, qbase, p.Q.rs, qstart + perm23(l31), p.Sq, p.D, hi);
  load_rows<KS>(docur, dobase, p.dO.rs, qstart + perm23(l31), p.Sq, p.D, hi);
  for (int q0 = qstart; q0 < p.Sq; q0 += 32) {
    bf16x8 dotf[MT][2], qtf[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bf16_t* dot = dotbase + (long long)(mt * 32 + l31) * p.dOt.ds + q0 + 8 * hi;
      const bf16_t* qt = qtbase + (long long)(mt * 32 + l31) * p.Qt.ds + q0 + 8 * hi;
      dotf[mt][0] = ld8(dot); dotf[mt][1] = ld8(dot + 16);
      qtf[mt][0] = ld8(qt); qtf[mt][1] = ld8(qt + 16);
    }
    float lse[16], del[16];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int qq = q0 + 16 * c + 8 * hi;  // Sqpad is a multiple of 32 -> in-bounds
      const float4 a0 = *reinterpret_cast<const float4*>(lsebase + qq), a1 = *reinterpret_cast<const float4*>(lsebase + qq + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(delbase + qq), b1 = *reinterpret_cast<const float4*>(delbase + qq + 4);
      lse[8 * c + 0] = a0.x; lse[8 * c + 1] = a0.y; lse[8 * c + 2] = a0.z; lse[8 * c + 3] = a0.w;
      lse[8 * c + 4] = a1.x; lse[8 * c + 5] = a1.y; lse[8 * c + 6] = a1.z; lse[8 * c + 7] = a1.w;
      del[8 * c + 0] = b0.x; del[8 * c + 1] = b0.y; del[8 * c + 2] = b0.z; del[8 * c + 3] = b0.w;
      del[8 * c + 4] = b1.x; del[8 * c + 5] = b1.y; del[8 * c + 6] = b1.z; del[8 * c + 7] = b1.w;
    }
    bf16x8 qnext[KS], donext[KS];
    load_rows<KS>(qnext, qbase, p.Q.rs, q0 + 32 + perm23(l31), p.Sq, p.D, hi);
    load_rows<KS>(donext, dobase, p.dO.rs, q0 + 32 + perm23(l31), p.Sq, p.D, hi);
    f32x16 sacc, dpacc;
    zero16(sacc);
    zero16(dpacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qcur[s], kf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(docur[s], vf[s], dpacc, 0, 0, 0);
    }
    // lane (key, hi), register r  <->  query q0 + 16*(r>>3) + 8*hi + (r&7)
    float sv[16];
    if (LUT) {
      if (kb0 - (q0 + 31) >= 128 || (q0 - (kb0 + 31)) >= 128) {
        const float bconst = lut[kb0 > q0 ? 256 : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bconst;
      } else {
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rel = key - (q0 + 16 * (r >> 3) + 8 * hi + (r & 7));
          bias[r] = lut[max(-128, min(128, rel)) + 128];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bias[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale;
    }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + 16 * (r >> 3) + 8 * hi + (r & 7);
      bool ok = key_ok && qq < p.Sq;
      if (CAUSAL) ok = ok && (key <= qq);
      const float pr = ok ? __expf(sv[r] - lse[r]) : 0.f;
      float dpv = dpacc[r], prd = pr;
      if (DROP) {
        const uint32_t hsh = attn_drop_hash(bh_idx + (uint32_t)qq, key, (p.Sk + 1) >> 1, drop_seed, p.drop.site);
        const bool keep = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= p.drop.thresh24;
        dpv = keep ? dpv * p.drop.inv_keep : 0.f;
        prd = keep ? pr * p.drop.inv_keep : 0.f;
      }
      pd[r] = prd;
      ds[r] = ok ? pr * (dpv - del[r]) * p.scale : 0.f;
    }
    const bf16x8 p0 = pack8(pd), p1 = pack8(pd + 8), s0 = pack8(ds), s1 = pack8(ds + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[mt][0], p0, dv[mt], 0, 0, 0);
      dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[mt][1], p1, dv[mt], 0, 0, 0);
      dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[mt][0], s0, dk[mt], 0, 0, 0);
      dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[mt][1], s1, dk[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { qcur[s] = qnext[s]; docur[s] = donext[s]; }
  }
  if (key < p.Sk) {
    bf16_t* kp = const_cast<bf16_t*>(p.dK.ptr) + b * p.dK.bs + h * p.dK.hs + (long long)key * p.dK.rs;
    bf16_t* vp = const_cast<bf16_t*>(p.dV.ptr) + b * p.dV.bs + h * p.dV.hs + (long long)key * p.dV.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D) {
          *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dk[mt][4 * g], dk[mt][4 * g + 1]), pack2bf(dk[mt][4 * g + 2], dk[mt][4 * g + 3]));
          *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dv[mt][4 * g], dv[mt][4 * g + 1]), pack2bf(dv[mt][4 * g + 2], dv[mt][4 * g + 3]));
        }
      }
  }
}

// ---- dK/dV with the query-side tiles shared through LDS (head_dim 64).  The four key-waves of a block consume the SAME Q, dO,
// Q^T, dO^T, LSE, Delta tile per step, so the block loads each tile once (16-B coalesced global loads -> registers -> LDS,
// issued one step ahead: the loads of tile t+1 fly while tile t is computed) instead of four times from L2; fragments are read
// with ds_read_b128 from XOR-swizzled rows (128-B rows: chunk ^= (row>>1)&7; 64-B rows: chunk ^= (row>>2)&3), conflict-free.
template <int FLAGS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_lds_kernel(const AttnArgs p) {
  constexpr int KS = 4, MT = 2;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP;
  constexpr int TILE = 4 * 4096 + 256;  // Q rows | dO rows | Q^T | dO^T | lse[32] delta[32]
  __shared__ float lut[LUT ? 257 : 1];
  __shared__ __attribute__((aligned(16))) char sm[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = tid; i < 257; i += 256) lut[i] = p.lut[h * 257 + i];
  }
  const int kb0 = (blockIdx.x * 4 + w) * 32;
  const int key = kb0 + l31;
  bool key_ok = key < p.Sk;
  bf16x8 kf[KS], vf[KS];
  load_rows<KS>(kf, p.K.ptr + b * p.K.bs + h * p.K.hs, p.K.rs, key, p.Sk, p.D, hi);
  load_rows<KS>(vf, p.V.ptr + b * p.V.bs + h * p.V.hs, p.V.rs, key, p.Sk, p.D, hi);
  if (MASK) key_ok = key_ok && (p.kmask[(long long)b * p.Skpad + min(key, p.Skpad - 1)] != 0);
  f32x16 dk[MT], dv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { zero16(dk[mt]); zero16(dv[mt]); }
  const bf16_t* qbase = p.Q.ptr + b * p.Q.bs + h * p.Q.hs;
  const bf16_t* dobase = p.dO.ptr + b * p.dO.bs + h * p.dO.hs;
  const bf16_t* qtbase = p.Qt.ptr + b * p.Qt.bs + h * p.Qt.hs;
  const bf16_t* dotbase = p.dOt.ptr + b * p.dOt.bs + h * p.dOt.hs;
  const float* lsebase = p.LSE + ((long long)(b * p.H + h)) * p.Sqpad;
  const float* delbase = p.Delta + ((long long)(b * p.H + h)) * p.Sqpad;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t bh_idx = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq;
  const int qstart = CAUSAL ? blockIdx.x * 128 : 0;  // block-uniform (the per-wave causal limit is applied by the mask)

  // staging roles of this thread: row tile (32 rows x 8 chunks) and transposed tile (64 rows x 4 chunks)
  const int r_row = tid >> 3, r_chunk = tid & 7, t_row = tid >> 2, t_chunk = tid & 3;
  const int r_off = r_row * 128 + ((r_chunk ^ ((r_row >> 1) & 7)) << 4);
  const int t_off = t_row * 64 + ((t_chunk ^ ((t_row >> 2) & 3)) << 4);
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 g_q, g_do, g_qt, g_dot;
  float4 g_st;
  auto gload = [&](int q0) {
    const int qr = q0 + r_row;
    const bool ok = qr < p.Sq && r_chunk * 8 < p.D;
    const long long ro = (long long)min(qr, p.Sq - 1) * p.Q.rs + min(r_chunk * 8, p.D - 8);
    const long long rdo = (long long)min(qr, p.Sq - 1) * p.dO.rs + min(r_chunk * 8, p.D - 8);
    g_q = sel8(ok, ld8(qbase + ro));
    g_do = sel8(ok, ld8(dobase + rdo));
    g_qt = ld8(qtbase + (long long)t_row * p.Qt.ds + q0 + t_chunk * 8);      // padded copies: always in bounds, zeros outside
    g_dot = ld8(dotbase + (long long)t_row * p.dOt.ds + q0 + t_chunk * 8);
    if (tid < 16) g_st = *reinterpret_cast<const float4*>((tid < 8 ? lsebase : delbase) + q0 + (tid & 7) * 4);
  };
  auto lstore = [&](int buf) {
    char* base = sm + buf * TILE;
    *reinterpret_cast<bf16x8*>(base + r_off) = g_q;
    *reinterpret_cast<bf16x8*>(base + 4096 + r_off) = g_do;
    *reinterpret_cast<bf16x8*>(base + 8192 + t_off) = g_qt;
    *reinterpret_cast<bf16x8*>(base + 12288 + t_off) = g_dot;
    if (tid < 16) *reinterpret_cast<float4*>(base + 16384 + tid * 16) = g_st;
  };
  gload(qstart);
  lstore(0);
  __syncthreads();

  const int frow = perm23(l31);                    // fragment row of the row tiles (the MFMA row permutation)
  const int f_sw = (frow >> 1) & 7, t_sw = (l31 >> 2) & 3;
  int it = 0;
  for (int q0 = qstart; q0 < p.Sq; q0 += 32, ++it) {
    const char* base = sm + (it & 1) * TILE;
    const bool more = q0 + 32 < p.Sq;
    if (more) gload(q0 + 32);
    bf16x8 qcur[KS], docur[KS], dotf[MT][2], qtf[MT][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int off = frow * 128 + (((2 * s + hi) ^ f_sw) << 4);
      qcur[s] = *reinterpret_cast<const bf16x8*>(base + off);
      docur[s] = *reinterpret_cast<const bf16x8*>(base + 4096 + off);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int off = (mt * 32 + l31) * 64 + (((2 * s2 + hi) ^ t_sw) << 4);
        qtf[mt][s2] = *reinterpret_cast<const bf16x8*>(base + 8192 + off);
        dotf[mt][s2] = *reinterpret_cast<const bf16x8*>(base + 12288 + off);
      }
    float lse[16], del[16];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* sp = reinterpret_cast<const float*>(base + 16384) + 16 * c + 8 * hi;
      const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sp + 32), b1 = *reinterpret_cast<const float4*>(sp + 36);
      lse[8 * c + 0] = a0.x; lse[8 * c + 1] = a0.y; lse[8 * c + 2] = a0.z; lse[8 * c + 3] = a0.w;
      lse[8 * c + 4] = a1.x; lse[8 * c + 5] = a1.y; lse[8 * c + 6] = a1.z; lse[8 * c + 7] = a1.w;
      del[8 * c + 0] = b0.x; del[8 * c + 1] = b0.y; del[8 * c + 2] = b0.z; del[8 * c + 3] = b0.w;
      del[8 * c + 4] = b1.x; del[8 * c + 5] = b1.y; del[8 * c + 6] = b1.z; del[8 * c + 7] = b1.w;
    }
    f32x16 sacc, dpacc;
    zero16(sacc);
    zero16(dpacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qcur[s], kf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(docur[s], vf[s], dpacc, 0, 0, 0);
    }
    float sv[16];
    if (LUT) {
      if (kb0 - (q0 + 31) >= 128 || (q0 - (kb0 + 31)) >= 128) {
        const float bconst = lut[kb0 > q0 ? 256 : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bconst;
      } else {
        float bias[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rel = key - (q0 + 16 * (r >> 3) + 8 * hi + (r & 7));
          bias[r] = lut[max(-128, min(128, rel)) + 128];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale + bias[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * p.scale;
    }
    float pd[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + 16 * (r >> 3) + 8 * hi + (r & 7);
      bool ok = key_ok && qq < p.Sq;
      if (CAUSAL) ok = ok && (key <= qq);
      const float pr = ok ? __expf(sv[r] - lse[r]) : 0.f;
      float dpv = dpacc[r], prd = pr;
      if (DROP) {
        const uint32_t hsh = attn_drop_hash(bh_idx + (uint32_t)qq, key, (p.Sk + 1) >> 1, drop_seed, p.drop.site);
        const bool keep = ((key & 1) ? (hsh >> 16) : (hsh & 0xffffu)) >= p.drop.thresh24;
        dpv = keep ? dpv * p.drop.inv_keep : 0.f;
        prd = keep ? pr * p.drop.inv_keep : 0.f;
      }
      pd[r] = prd;
      ds[r] = ok ? pr * (dpv - del[r]) * p.scale : 0.f;
    }
    const bf16x8 p0 = pack8(pd), p1 = pack8(pd + 8), s0 = pack8(ds), s1 = pack8(ds + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[mt][0], p0, dv[mt], 0, 0, 0);
      dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[mt][1], p1, dv[mt], 0, 0, 0);
      dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[mt][0], s0, dk[mt], 0, 0, 0);
      dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[mt][1], s1, dk[mt], 0, 0, 0);
    }
    if (more) lstore((it & 1) ^ 1);
    __syncthreads();
  }
  if (key < p.Sk) {
    bf16_t* kp = const_cast<bf16_t*>(p.dK.ptr) + b * p.dK.bs + h * p.dK.hs + (long long)key * p.dK.rs;
    bf16_t* vp = const_cast<bf16_t*>(p.dV.ptr) + b * p.dV.bs + h * p.dV.hs + (long long)key * p.dV.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D) {
          *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dk[mt][4 * g], dk[mt][4 * g + 1]), pack2bf(dk[mt][4 * g + 2], dk[mt][4 * g + 3]));
          *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dv[mt][4 * g], dv[mt][4 * g + 1]), pack2bf(dv[mt][4 * g + 2], dv[mt][4 * g + 3]));
        }
      }
  }
}

// ---- [B,S,H,D]-strided rows -> [B,H,DP,Spad] transposed, zero padded (Spad multiple of 32, DP multiple of 32).
// 16-B loads along d, LDS tile, 16-B stores along s.  Optional dropout on the SOURCE elements (index = row * ncols + col of
// the [B*S, H*D] matrix) so that drop(x)^T for the LoRA dA GEMM never has to be materialised un-transposed.
__global__ __launch_bounds__(256) void head_transpose_kernel(T4 src, bf16_t* dst, int S, int D, int DP, int Spad, DropoutArg drop) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[32][96 + 8];
  const int b = blockIdx.z, h = blockIdx.y, s0 = blockIdx.x * 32;
  const bf16_t* sp = src.ptr + b * src.bs + h * src.hs;
  const int cpr = DP / 8;  // 16-B chunks per row
  const uint32_t seed = drop.seed_ptr ? *drop.seed_ptr : 0u;
  const int ncols = gridDim.y * D;
  for (int i = threadIdx.x; i < 32 * cpr; i += 256) {
    const int r = i / cpr, c8 = (i % cpr) * 8;
    const int sidx = s0 + r;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (sidx < S && c8 < D) {
      v = *reinterpret_cast<const bf16x8*>(sp + (long long)sidx * src.rs + c8);
      if (drop.seed_ptr) {
        const uint32_t base = (uint32_t)(b * S + sidx) * (uint32_t)ncols + (uint32_t)(h * D + c8);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = mrb_keep(base + j, seed, drop.site, drop.thresh24) ? (short)f2bf(bf2f((bf16_t)v[j]) * drop.inv_keep) : (short)0;
      }
    }
    *reinterpret_cast<bf16x8*>(&tile[r][c8]) = v;
  }
  __syncthreads();
  bf16_t* dp = dst + ((long long)(b * gridDim.y + h) * DP) * Spad + s0;
  for (int i = threadIdx.x; i < DP * 4; i += 256) {
    const int d = i >> 2, q = (i & 3) * 8;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)tile[q + j][d];
    *reinterpret_cast<bf16x8*>(dp + (long long)d * Spad + q) = o;
  }
}

static int attn_fill(AttnArgs& a, const void* Q, const long long* qs, const void* K, const long long* ks, const void* V,
                     const long long* vs, int B, int H, int Sq, int Sk, int D) {
  MRB_REQUIRE(B > 0 && H > 0 && Sq > 0 && Sk > 0, "attention: empty problem");
  MRB_REQUIRE(D > 0 && D <= 96 && (D % 8) == 0, "attention: head_dim must be a multiple of 8 and <= 96 (got %d)", D);
  a.Q = T4{(const bf16_t*)Q, qs[0], qs[1], qs[2]};
  a.K = T4{(const bf16_t*)K, ks[0], ks[1], ks[2]};
  a.V = T4{(const bf16_t*)V, vs[0], vs[1], vs[2]};
  a.B = B; a.H = H; a.Sq = Sq; a.Sk = Sk; a.D = D; a.Sqpad = (Sq + 31) / 32 * 32; a.Skpad = (Sk + 31) / 32 * 32;
  return MRBLIP_OK;
}

static void attn_drop(AttnArgs& a, const uint32_t* seed_ptr, uint32_t site, float p_drop) {
  a.drop.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  a.drop.site = site;
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  a.drop.inv_keep = 1.0f / (1.0f - p_drop);
}

static int attn_flags(const AttnArgs& a, int causal) {
  return (a.lut ? F_LUT : 0) | (a.kmask ? F_MASK : 0) | (causal ? F_CAUSAL : 0) | (a.drop.seed_ptr ? F_DROP : 0);
}

// The flag combinations the hot path uses (anything else is rejected loudly): 0 plain (ViT, eval Q-Former), DROP (Q-Former),
// LUT|MASK[|DROP] (T5 encoder), LUT|MASK|CAUSAL[|DROP] (decoder self), MASK[|DROP] (decoder cross).
#define ATTN_FLAG_CASES(X, DPV)                                                                                          \
  X(DPV, 0) X(DPV, F_DROP) X(DPV, F_LUT | F_MASK) X(DPV, F_LUT | F_MASK | F_DROP) X(DPV, F_LUT | F_MASK | F_CAUSAL)       \
  X(DPV, F_LUT | F_MASK | F_CAUSAL | F_DROP) X(DPV, F_MASK) X(DPV, F_MASK | F_DROP)                                    \
  X(DPV, F_LUT) X(DPV, F_LUT | F_DROP) X(DPV, F_LUT | F_CAUSAL) X(DPV, F_LUT | F_CAUSAL | F_DROP)

template <int DP>
static int launch_fwd(const AttnArgs& a, int flags, hipStream_t stream) {
  const bool split = a.Sq <= 32 && a.Sk >= 256 && !(flags & F_CAUSAL);
  dim3 grid(split ? 1 : (a.Sq + 127) / 128, a.H, a.B);
#define X(DPV, FL)                                                                                             \
  if (flags == (FL)) {                                                                                         \
    if (split && !((FL) & F_CAUSAL)) hipLaunchKernelGGL((attn_fwd_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT>), grid, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((attn_fwd_kernel<DPV, (FL)>), grid, dim3(256), 0, stream, a);                       \
    return mrblip_check_launch("attention_fwd");                                                               \
  }
  ATTN_FLAG_CASES(X, DP)
#undef X
  mrblip_set_error("attention_fwd: unsupported feature combination (flags=%d)", flags);
  return MRBLIP_EINVAL;
}

template <int DP>
static int launch_bwd(const AttnArgs& a, int flags, hipStream_t stream) {
  const bool split = a.Sq <= 32 && a.Sk >= 256 && !(flags & F_CAUSAL);
  dim3 gq(split ? 1 : (a.Sq + 127) / 128, a.H, a.B), gk((a.Sk + 127) / 128, a.H, a.B);
#define X(DPV, FL)                                                                                             \
  if (flags == (FL)) {                                                                                         \
    if (split && !((FL) & F_CAUSAL)) hipLaunchKernelGGL((attn_bwd_dq_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT>), gq, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<DPV, (FL)>), gq, dim3(256), 0, stream, a);                      \
    if (DPV == 64 && a.D == 64 && a.Sq > 32) hipLaunchKernelGGL((attn_bwd_dkv_lds_kernel<(FL)>), gk, dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DPV, (FL)>), gk, dim3(256), 0, stream, a);                     \
    return mrblip_check_launch("attention_bwd");                                                               \
  }
  ATTN_FLAG_CASES(X, DP)
#undef X
  mrblip_set_error("attention_bwd: unsupported feature combination (flags=%d)", flags);
  return MRBLIP_EINVAL;
}

// strides arrays: {batch, head, row} in elements.  Vt: [B,H,DP,Skpad] with DP = roundup32(D), Skpad = roundup32(Sk).
// kmask (optional): int32 [B, Skpad] (rows padded to a multiple of 32 entries).
extern "C" int mrblip_attention_fwd(const void* Q, const long long* q_strides, const void* K, const long long* k_strides,
                                    const void* Vt, void* O, const long long* o_strides, float* LSE, int B, int H, int Sq, int Sk,
                                    int D, float scale, const float* bias_lut, const int* kmask, int causal,
                                    const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  AttnArgs a = {};
  long long dummy[3] = {0, 0, 0};
  if (int e = attn_fill(a, Q, q_strides, K, k_strides, nullptr, dummy, B, H, Sq, Sk, D)) return e;
  const int DP = (D + 31) / 32 * 32, Skpad = a.Skpad;
  a.Vt = T4T{(const bf16_t*)Vt, (long long)H * DP * Skpad, (long long)DP * Skpad, Skpad};
  a.O = T4{(const bf16_t*)O, o_strides[0], o_strides[1], o_strides[2]};
  a.LSE = LSE; a.lut = bias_lut; a.kmask = kmask; a.scale = scale;
  attn_drop(a, seed_ptr, site, p_drop);
  const int flags = attn_flags(a, causal);
  if (DP == 32) return launch_fwd<32>(a, flags, stream);
  if (DP == 64) return launch_fwd<64>(a, flags, stream);
  MRB_REQUIRE(flags == 0, "attention_fwd: head_dim > 64 supports the plain (ViT) form only");
  hipLaunchKernelGGL((attn_fwd_kernel<96, 0>), dim3((Sq + 127) / 128, H, B), dim3(256), 0, stream, a);
  return mrblip_check_launch("attention_fwd");
}

// Backward.  Needs the forward's O and LSE, the transposed copies Kt [B,H,DP,Skpad], Qt / dOt [B,H,DP,Sqpad];
// writes Delta [B,H,Sqpad] (scratch), dQ, dK, dV (bf16, strided like their forward tensors).
extern "C" int mrblip_attention_bwd(const void* Q, const long long* q_strides, const void* K, const long long* k_strides,
                                    const void* V, const long long* v_strides, const void* O, const long long* o_strides,
                                    const void* dO, const long long* do_strides, const void* Kt, const void* Qt, const void* dOt,
                                    const float* LSE, float* Delta, void* dQ, const long long* dq_strides, void* dK,
                                    const long long* dk_strides, void* dV, const long long* dv_strides, int B, int H, int Sq, int Sk,
                                    int D, float scale, const float* bias_lut, const int* kmask, int causal,
                                    const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  AttnArgs a = {};
  if (int e = attn_fill(a, Q, q_strides, K, k_strides, V, v_strides, B, H, Sq, Sk, D)) return e;
  MRB_REQUIRE(D <= 64, "attention_bwd: head_dim <= 64 only (the ViT is frozen, its attention needs no backward)");
  const int DP = (D + 31) / 32 * 32, Skpad = a.Skpad, Sqpad = a.Sqpad;
  a.O = T4{(const bf16_t*)O, o_strides[0], o_strides[1], o_strides[2]};
  a.dO = T4{(const bf16_t*)dO, do_strides[0], do_strides[1], do_strides[2]};
  a.dQ = T4{(const bf16_t*)dQ, dq_strides[0], dq_strides[1], dq_strides[2]};
  a.dK = T4{(const bf16_t*)dK, dk_strides[0], dk_strides[1], dk_strides[2]};
  a.dV = T4{(const bf16_t*)dV, dv_strides[0], dv_strides[1], dv_strides[2]};
  a.Kt = T4T{(const bf16_t*)Kt, (long long)H * DP * Skpad, (long long)DP * Skpad, Skpad};
  a.Qt = T4T{(const bf16_t*)Qt, (long long)H * DP * Sqpad, (long long)DP * Sqpad, Sqpad};
  a.dOt = T4T{(const bf16_t*)dOt, (long long)H * DP * Sqpad, (long long)DP * Sqpad, Sqpad};
  a.LSE = const_cast<float*>(LSE); a.Delta = Delta; a.lut = bias_lut; a.kmask = kmask; a.scale = scale;
  attn_drop(a, seed_ptr, site, p_drop);
  const int flags = attn_flags(a, causal);
  if (DP == 32) return launch_bwd<32>(a, flags, stream);
  return launch_bwd<64>(a, flags, stream);
}

extern "C" int mrblip_head_transpose(const void* src, const long long* strides, void* dst, int B, int H, int S, int D, int Spad,
                                     const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  MRB_REQUIRE(B > 0 && H > 0 && S > 0 && D > 0 && D <= 96, "head_transpose: bad shape");
  const int DP = (D + 31) / 32 * 32;
  if (Spad <= 0) Spad = (S + 31) / 32 * 32;
  MRB_REQUIRE(Spad >= S && (Spad % 32) == 0, "head_transpose: Spad must be a multiple of 32 and >= S");
  T4 s{(const bf16_t*)src, strides[0], strides[1], strides[2]};
  MRB_REQUIRE((D % 8) == 0 && (strides[2] % 8) == 0 && ((uintptr_t)src % 16) == 0, "head_transpose: rows must be 16-B aligned, D %% 8 == 0");
  AttnArgs tmp = {};
  attn_drop(tmp, seed_ptr, site, p_drop);
  hipLaunchKernelGGL(head_transpose_kernel, dim3(Spad / 32, H, B), dim3(256), 0, stream, s, (bf16_t*)dst, S, D, DP, Spad, tmp.drop);
  return mrblip_check_launch("head_transpose");
}
