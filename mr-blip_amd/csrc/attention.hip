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
  uint32_t* dbits;     // optional [B*H, Skpad/32, Sqpad]: keep bits of the probability dropout (bit i of word (kt, q) <-> key 32 kt + i),
                       // written by the LDS forward kernel and read back by the LDS backward kernels instead of re-hashing
  // F_XS (round 4): the key range of a (batch, head) split over xs_n BLOCKS (gridDim.x); partial results meet in xs_ws, the last arriver
  // (ticket xs_cnt[b * H + h], zero before the launch and reset by the merger) combines them in chunk order
  float* xs_ws; uint32_t* xs_cnt; int xs_n;
};

enum { F_LUT = 1, F_MASK = 2, F_CAUSAL = 4, F_DROP = 8, F_SPLIT = 16, F_DBITS = 32, F_VROW = 64, F_KS2 = 128,
       F_F16 = 256, F_XS = 512 };   // F_F16 (round 4): Q / K / V / O and the probabilities fed to the second product are IEEE fp16 (the fp16-operand ViT)

// Attention-probability dropout draws (v3, round 5).  ONE 32-bit hash serves a key QUAD: index = row * ceil(Sk / 4) + key / 4 with
// row = (b * H + h) * Sq + q, hash h = mrb_lin_fin24(index * MRB_H1 + mrb_lin_base(seed, site)) (common.h: the finaliser runs on the
// full-rate 24-bit multiplier), g = rotr(h, 8); key 4i + j takes the 16-bit draw  j = 0: h & 0xffff,  1: h >> 16,  2: g & 0xffff,
// 3: g >> 16  — every byte of h is the TOP byte of exactly one draw; keep iff draw >= round(p * 65536) (p = 0.1 -> 0.10001, the element
// dropout's resolution; v2 had 11-bit windows: 0.1001).  A draw's low byte is another draw's top byte, which matters only when the
// top byte equals the threshold's: neighbour conditionals 0.100-0.103 (v2: 0.1016), measured on the restatement
// (tests/test_host_cpu.py).  WHY this shape: the words h and g hold the draws of the element pairs (0, 1) and (2, 3) in their 16-bit
// halves, exactly where a packed bf16 pair holds the two probabilities — the LDS forward applies the mask to the PACKED pair with
// three packed-16-bit instructions (saturating subtract, min, multiply) and collects the keep bits with one shift-or per pair, instead
// of extract + compare + select + select per ELEMENT (v2: ~110 VALU per 32 x 32 score tile for the dropout alone).
#define NEG_BIG (-1.0e30f)
__device__ __forceinline__ uint32_t attn_rot8(uint32_t h) { return __builtin_amdgcn_alignbit(h, h, 8); }
__device__ __forceinline__ uint32_t attn_draw(uint32_t hash, int j) {   // scalar form (per-wave kernels, key-owner backward)
  const uint32_t w = (j & 2) ? attn_rot8(hash) : hash;
  return (j & 1) ? (w >> 16) : (w & 0xffffu);
}
// packed form: halves of w = two draws; returns 1 / 0 in each 16-bit half (keep / drop).  tm1x2 = (thresh - 1) in both halves, thresh >= 1.
// (inline asm: written with vector builtins, LLVM canonicalises min(sub_sat(x, t), 1) to zext(x > t) and emits compare + select + byte
// permute per element again)
__device__ __forceinline__ uint32_t attn_keep01(uint32_t w, uint32_t tm1x2) {
  uint32_t s;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(s) : "v"(w), "s"(tm1x2));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(s) : "v"(s), "s"(0x00010001u));
  return s;
}
__device__ __forceinline__ uint32_t attn_pkmul(uint32_t pair, uint32_t k01) {   // a packed 16-bit pair times its 0 / 1 keep factors
  uint32_t r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(pair), "v"(k01));
  return r;
}
// lane <-> lane ^ 32 exchange without the LDS crossbar: v_permlane32_swap (gfx950) on two copies; {r[0], r[1]} = {own or partner, partner
// or own} depending on the half, so any SYMMETRIC combination of the two is what __shfl_xor(x, 32) + combine gave
__device__ __forceinline__ float attn_max_x32(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
}
__device__ __forceinline__ uint32_t attn_or_x32(uint32_t x) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return (uint32_t)r[0] | (uint32_t)r[1];
}
__device__ __forceinline__ float attn_sum_x32(float x) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
}
// a wave-uniform float the compiler must keep as ONE finished scalar: the value passes through a VGPR it cannot see into, so the
// arithmetic that produced it (a scalar load times a constant: the scalar unit has no float multiply) is not re-done at every use
__device__ __forceinline__ float attn_uniform(float x) {
  asm volatile("" : "+v"(x));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
// keep-bit word of a (32-key tile, query): the forward collects the bit of register pair i (registers 2i, 2i + 1 of lane half hi) at
// positions 8 hi + i and 16 + 8 hi + i, i.e. key 16 c + 8 hi + j (register r = 8 c + j) sits at bit 16 (j & 1) + 8 hi + 4 c + (j >> 1).
// Any fixed permutation serves the backward kernels equally (one v_bfe_i32 per element either way).
__device__ __forceinline__ int attn_bitpos_reg(int r) { return 16 * (r & 1) + (r >> 1); }   // of register r, in the word shifted right by 8 hi
__device__ __forceinline__ int attn_bitpos_key(int kk) {                                      // of key kk = 0..31 of the tile
  return 16 * (kk & 1) + 8 * ((kk >> 3) & 1) + 4 * (kk >> 4) + ((kk & 7) >> 1);
}
// key-owner (dK/dV) kernels: lane owns one key, register j a query row; no sharing (these forms only run where the forward's keep bits
// are not stored: the 32-query Q-Former and the 12-token decoder)
__device__ __forceinline__ void drop_draws8_keyowner(uint32_t (&draw)[8], uint32_t rowbase, int key, int skq, uint32_t dbase) {
#pragma unroll
  for (int j = 0; j < 8; ++j)
    draw[j] = attn_draw(mrb_lin_fin24(((rowbase + (uint32_t)j) * (uint32_t)skq + (uint32_t)(key >> 2)) * MRB_H1 + dbase), key & 3);
}
// query-owner kernels (forward, dQ): lane (q, hi) owns keys k0 + 16c + 8hi + j (c < 2, j < 8) of a 32-key tile.  t_lane =
// (row * skq + 2 * hi) * MRB_H1 + dbase is a lane constant; the tile term ((k0 >> 2) + 4c + qd) * MRB_H1 is wave-uniform.
// keep bit of register r = 8c + 4qd + j  ->  apply(r, keep).
template <typename F>
__device__ __forceinline__ void attn_keep16(uint32_t t_lane, int k0, uint32_t thresh, F&& apply) {  // apply(r, keep) for r = 0..15
#pragma unroll
  for (int cq = 0; cq < 4; ++cq) {  // cq = 2c + qd
    const uint32_t hsh = mrb_lin_fin24(t_lane + (uint32_t)((k0 >> 2) + 4 * (cq >> 1) + (cq & 1)) * MRB_H1);
#pragma unroll
    for (int j = 0; j < 4; ++j) apply(4 * cq + j, attn_draw(hsh, j) >= thresh);
  }
}

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
template <bool F16>
__device__ __forceinline__ bf16x8 pack8x(const float* v) {
  union { bf16x8 v8; uint32_t u[4]; } r;
  r.u[0] = pack2x<F16>(v[0], v[1]); r.u[1] = pack2x<F16>(v[2], v[3]); r.u[2] = pack2x<F16>(v[4], v[5]); r.u[3] = pack2x<F16>(v[6], v[7]);
  return r.v8;
}
// Scores live in the log2 domain: s2 = (q.k * scale + bias) * log2(e), so every softmax exponential is one v_exp_f32
// (no multiply), and masked scores are -inf (exp2 -> exact 0 without a select; running maxima start at the finite NEG_BIG).
#define MRB_LOG2E 1.4426950408889634f
#define MRB_LN2 0.6931471805599453f
#define NEG_INF (-__builtin_inff())
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
template <bool V> struct BoolC { static constexpr bool value = V; };

// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2: with (q-block, head, batch) taken
// straight from blockIdx the q-blocks that share one head's K / V tiles land on DIFFERENT XCDs and each of them pulls the tiles through
// its own L2.  This bijective remap (the tile GEMM's) gives XCD x the x-th contiguous eighth of the linear work list instead, so the
// q-blocks of a head (and the heads of a frame) are neighbours in one L2.  MRB_ATTN_NO_XCD_REMAP (build flag) = identity, for A/B.
__device__ __forceinline__ void attn_block(int& bx, int& by, int& bz) {
#ifdef MRB_ATTN_NO_XCD_REMAP
  bx = blockIdx.x; by = blockIdx.y; bz = blockIdx.z;
#else
  const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
  int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const int q = total >> 3, r = total & 7, xcd = id & 7, idx = id >> 3;
  id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bx = id % gx; by = (id / gx) % gy; bz = id / (gx * gy);
#endif
}

// sv[r] = sacc[r] * scale2 + lut[clamp(rel_r)] with rel_r = relbase + SGN * (16*(r>>3) + (r&7)); a tile whose every |rel| >= 128
// shares one bucket (far_idx) -> one LUT read instead of sixteen.
template <bool LUT, int SGN>
__device__ __forceinline__ void tile_scores(float (&sv)[16], const f32x16& sacc, float scale2, const float* lut, bool far_tile, int far_idx, int relbase) {
  if (LUT) {
    if (far_tile) {
      // wave-uniform bucket: through the scalar unit, so that each score is ONE v_fma with the addend in an SGPR — left in a VGPR the
      // compiler merged this arm with the near arm's fmac and paid 15 broadcast moves per tile
      const float bconst = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lut[far_idx])));
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = __builtin_fmaf(sacc[r], scale2, bconst);
    } else {
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rel = relbase + SGN * (16 * (r >> 3) + (r & 7));
        bias[r] = lut[max(-128, min(128, rel)) + 128];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * scale2 + bias[r];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) sv[r] = sacc[r] * scale2;
  }
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

// LDS kernels (round 3): the key mask as a BITMAP in LDS, word t = keys 32t .. 32t + 31, built once per block — per 32-key tile a lane
// then needs one broadcast LDS read and two bit-field ops instead of four 16-B global loads and 32 compare / shift / or instructions.
#define ATTN_MASK_WORDS 256   // 8192 keys
__device__ __forceinline__ void attn_mask_bitmap(uint32_t* mb, const int* km, int skpad, int tid, int nt) {
  for (int wd = tid; wd < (skpad >> 5); wd += nt) {
    uint32_t bits = 0;
    const int4* p4 = reinterpret_cast<const int4*>(km + 32 * wd);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int4 a = p4[j];
      bits |= ((uint32_t)(a.x != 0) | ((uint32_t)(a.y != 0) << 1) | ((uint32_t)(a.z != 0) << 2) | ((uint32_t)(a.w != 0) << 3)) << (4 * j);
    }
    mb[wd] = bits;
  }
}
// this lane's 16 valid-key bits of tile t (register order: bit 8c + j <-> key 32t + 16c + 8hi + j), as mask_bits() returns them
__device__ __forceinline__ uint32_t mask_bits_lds(const uint32_t* mb, int t, int hi) {
  const uint32_t wd = mb[t] >> (8 * hi);
  return (wd & 0xffu) | ((wd >> 8) & 0xff00u);
}


// ---- F_XS: few queries (<= 32: the decoder's cross attention, 8-14 label rows against 2012 keys) leave ONE block per head in the key-split
// form above — 32 blocks on a 256-CU chip, each streaming 515 KB of K / V through one CU as a chain of dependent tiles (25 us forward, 37 us
// dQ per decoder layer).  With F_XS the key range is cut into gridDim.x chunks of whole 32-key tiles, one block each; a block's four waves
// split ITS chunk as before and wave 0 publishes the block's partial — forward: (m, l, O), dQ: the partial sum — to a workspace with 16-B
// WRITE-THROUGH (sc1) stores, drains them, and takes a ticket (relaxed agent-scope atomic).  The block that draws the last ticket resets it,
// does ONE agent-scope acquire and combines the gridDim.x partials in CHUNK order, whichever block it is: deterministic, and independent
// of dispatch order and XCD placement (cdna_hip_programming.md Guideline 16, form R1 with the ticket as the flag).
typedef uint32_t attn_u32x4 __attribute__((ext_vector_type(4)));
template <int NV4>
__device__ __forceinline__ void xs_publish(float* slot, const f32x4 (&v)[NV4], int lane) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(slot, 0, NV4 * 64 * 16, 0x00020000);
#pragma unroll
  for (int g = 0; g < NV4; ++g)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(attn_u32x4, v[g]), r, (uint32_t)((g * 64 + lane) * 16), 0, 16 /* sc1: write-through */);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the storing wave drains its stores before the ticket
}
template <int NV4>
__device__ __forceinline__ void xs_fetch(const float* slot, f32x4 (&v)[NV4], int lane) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slot), 0, NV4 * 64 * 16, 0x00020000);
#pragma unroll
  for (int g = 0; g < NV4; ++g) v[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)((g * 64 + lane) * 16), 0, 16 /* sc1 */));
}
// true for the one wave that merges (the caller is a single wave: the block's wave 0)
__device__ __forceinline__ bool xs_last(uint32_t* cnt, int n, int lane) {
  uint32_t old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
  if (old != (uint32_t)(n - 1)) return false;
  if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// NSW: waves that split the key range of the block's one query tile in the SPLIT form (4, or 8 for long key ranges: the decoder's cross
// attention has 8-14 queries against 2012 keys and ONE block per head — its waves walk 503 keys each, a chain of 16 dependent tiles)
template <int DP, int FLAGS, int NSW = 4>
__global__ __launch_bounds__(NSW * 64, NSW == 4 ? 2 : 1) void attn_fwd_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP, SPLIT = FLAGS & F_SPLIT;
  constexpr bool XS = FLAGS & F_XS;
  static_assert(NSW == 4 || SPLIT, "more than four waves only in the key-split form");
  static_assert(!XS || (SPLIT && !CAUSAL), "the cross-block key split extends the key-split form");
  __shared__ float lut[LUT ? 257 : 1];
  __shared__ float red[SPLIT ? (NSW - 1) * (MT * 16 + 2) * 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += NSW * 64) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
    __syncthreads();
  }
  const int q0 = XS ? 0 : SPLIT ? blockIdx.x * 32 : (blockIdx.x * 4 + w) * 32;
  if (!SPLIT && q0 >= p.Sq) return;
  const int q = q0 + l31;
  const bool q_ok = q < p.Sq;
  bf16x8 qf[KS];
  load_rows<KS>(qf, p.Q.ptr + b * p.Q.bs + h * p.Q.hs, p.Q.rs, q, p.Sq, p.D, hi);
  float m_run = NEG_BIG, l_run = 0.f;
  f32x16 o[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(o[mt]);
  // XS: this block's chunk of whole key tiles
  const int xs_tiles = XS ? ((p.Sk + 31) / 32 + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  const int kend = XS ? min(p.Sk, ((int)blockIdx.x + 1) * xs_tiles * 32) : CAUSAL ? min(p.Sk, q0 + 32) : p.Sk;
  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vtbase = p.Vt.ptr + b * p.Vt.bs + h * p.Vt.hs;
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const uint32_t t_lane = DROP ? (row_id * (uint32_t)((p.Sk + 3) >> 2) + 2u * (uint32_t)hi) * MRB_H1 + mrb_lin_base(drop_seed, p.drop.site) : 0u;
  const int kstart = (XS ? (int)blockIdx.x * xs_tiles * 32 : 0) + (SPLIT ? w * 32 : 0), kstep = SPLIT ? NSW * 32 : 32;
  const float scale2 = p.scale * MRB_LOG2E;

  // software pipeline: K fragments of the next tile and V^T fragments of this tile are in flight during the score math
  bf16x8 kcur[KS];
  load_rows<KS>(kcur, kbase, p.K.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  // EDGE tiles (ragged key tail, causal diagonal, key mask) pay for per-element validity; interior tiles do not.
  auto tile = [&](auto edge_c, int k0) {
    constexpr bool EDGE = decltype(edge_c)::value;
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
    tile_scores<LUT, 1>(sv, sacc, scale2, lut, k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128, k0 > q0 ? 256 : 0, k0 + 8 * hi - q);
    if (EDGE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
        bool ok = key < p.Sk && ((vmask >> r) & 1u);
        if (CAUSAL) ok = ok && (key <= q);
        sv[r] = ok ? sv[r] : NEG_INF;
      }
    }
    float mx = fmaxf(sv[0], sv[1]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sv[r]), sv[r + 1]);
    mx = attn_max_x32(mx);
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {  // lazy rescale: once the running max has settled nothing is multiplied
      const float m_new = fmaxf(m_run, mx);
      const float alpha = ex2(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    }
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = ex2(sv[r] - m_run);
      psum += pv[r];
    }
    if (DROP) {  // the 1/(1-p) factor is applied once to O at the end
      attn_keep16(t_lane, k0, p.drop.thresh24, [&](int r, bool kp) { pv[r] = kp ? pv[r] : 0.f; });
    }
    l_run += psum;
    const bf16x8 pf0 = pack8(pv), pf1 = pack8(pv + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[mt][0], pf0, o[mt], 0, 0, 0);
      o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[mt][1], pf1, o[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) kcur[s] = knext[s];
  };
  for (int k0 = kstart; k0 < kend; k0 += kstep) {
    const bool edge = MASK || k0 + 32 > p.Sk || (CAUSAL && k0 + 31 > q0);
    if (edge) tile(BoolC<true>(), k0);
    else tile(BoolC<false>(), k0);
  }
  float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (SPLIT) {  // merge the key-range partials: wave 0 collects (m, l, O) of waves 1 .. NSW - 1
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
    for (int j = 0; j < NSW - 1; ++j) m_all = fmaxf(m_all, red[j * STR + lane]);
    const float f0 = ex2(m_run - m_all);
    l_tot *= f0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[mt][i] *= f0;
#pragma unroll
    for (int j = 0; j < NSW - 1; ++j) {
      const float* r = red + j * STR;
      const float fj = ex2(r[lane] - m_all);
      l_tot += r[64 + lane] * fj;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[mt][i] += r[(2 + mt * 16 + i) * 64 + lane] * fj;
    }
    m_run = m_all;
  }
  if constexpr (XS) {   // (wave 0 only from here) publish (m, l, O) of this chunk; the last arriver combines all chunks in chunk order
    constexpr int NV4 = (2 + MT * 16 + 3) / 4;
    const int nx = (int)gridDim.x;
    float* base = p.xs_ws + (long long)(b * p.H + h) * nx * (NV4 * 64 * 4);
    f32x4 v[NV4];
    {
      float flat[NV4 * 4];
      flat[0] = m_run; flat[1] = l_tot;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) flat[2 + mt * 16 + i] = o[mt][i];
#pragma unroll
      for (int i = 2 + MT * 16; i < NV4 * 4; ++i) flat[i] = 0.f;
#pragma unroll
      for (int g = 0; g < NV4; ++g) v[g] = f32x4{flat[4 * g], flat[4 * g + 1], flat[4 * g + 2], flat[4 * g + 3]};
    }
    xs_publish<NV4>(base + (long long)blockIdx.x * (NV4 * 64 * 4), v, lane);
    if (!xs_last(p.xs_cnt + b * p.H + h, nx, lane)) return;
    float m_all = NEG_BIG;
    for (int j = 0; j < nx; ++j) {   // pass 1: the overall maximum
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base + (long long)j * (NV4 * 64 * 4), 0, NV4 * 64 * 16, 0x00020000);
      const f32x4 g0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)(lane * 16), 0, 16));
      m_all = fmaxf(m_all, g0[0]);
    }
    l_tot = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(o[mt]);
    for (int j = 0; j < nx; ++j) {   // pass 2: chunk order
      xs_fetch<NV4>(base + (long long)j * (NV4 * 64 * 4), v, lane);
      float flat[NV4 * 4];
#pragma unroll
      for (int g = 0; g < NV4; ++g) { flat[4 * g] = v[g][0]; flat[4 * g + 1] = v[g][1]; flat[4 * g + 2] = v[g][2]; flat[4 * g + 3] = v[g][3]; }
      const float fj = ex2(flat[0] - m_all);
      l_tot += flat[1] * fj;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[mt][i] += flat[2 + mt * 16 + i] * fj;
    }
    m_run = m_all;
  }
  const float inv = l_tot > 0.f ? (DROP ? p.drop.inv_keep : 1.0f) / l_tot : 0.f;
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
    if (p.LSE && hi == 0) p.LSE[((long long)(b * p.H + h)) * p.Sqpad + q] = m_run * MRB_LN2 + __logf(fmaxf(l_tot, 1e-37f));
  }
}

// ---- forward with the key-side tiles shared through LDS.  The four query-waves of a block consume the SAME K / V^T tiles, so the
// block fetches each 64-key stage once with LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip) instead of every wave
// gathering 32-B row pieces from L1/L2 (the per-wave form is bound by the texture-address path, not by MFMA or VALU).  Two stages,
// one barrier per stage: the DMA of stage t+1 flies while stage t is consumed.  LDS rows are XOR-swizzled at 16-B granularity through
// the SOURCE address (the DMA destination is lane-linear), reads are conflict-free ds_read_b128.
typedef __attribute__((address_space(3))) void* attn_lds_ptr_t;

// one stage of LDS-DMA: CNT_A + CNT_B instructions per thread, destinations lane-linear (16 B per lane, 1 KB per wave and instruction)
template <int CNT_A, int CNT_B, int NT = 256>
__device__ __forceinline__ void attn_stage_dma(char* dstA, char* dstB, const void* srcA, uint32_t bytesA, const void* srcB, uint32_t bytesB,
                                               const uint32_t* vA, const uint32_t* vB, int w, uint32_t sA, uint32_t sB) {
  // (the buffer resource type exists on the device side only: it cannot appear in a signature the host pass also parses)
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(srcA), 0, (int)bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(srcB), 0, (int)bytesB, 0x00020000);
#pragma unroll
  for (int j = 0; j < CNT_A; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (attn_lds_ptr_t)(dstA + (j * NT + w * 64) * 16), 16, vA[j], sA, 0, 0);
#pragma unroll
  for (int j = 0; j < CNT_B; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (attn_lds_ptr_t)(dstB + (j * NT + w * 64) * 16), 16, vB[j], sB, 0, 0);
}

#ifndef ATTN96_BLOCKS
#define ATTN96_BLOCKS 2   // resident blocks per CU the ViT form (DP = 96) is compiled for (3 = 168 VGPRs with 36 spilled dwords: measured, see DESIGN)
#endif
// F_KS2 (round 3, head_dim 64): EIGHT waves per block; waves w and w + 4 share one 32-row query tile and split the keys — wave w takes
// the first 32-key sub-tile of every 64-key stage, wave w + 4 the second — and merge their (m, l, O) through LDS at the end.  Same stages,
// same LDS traffic, but twice the waves: at S = 2012 x 32 heads there are only 2016 query tiles, i.e. TWO waves per SIMD, and a
// VALU-bound kernel with 2 waves per SIMD left the VALU pipe idle a third of the time (round-2 counters: 67 % busy).
template <int DP, int FLAGS>
__global__ __launch_bounds__((FLAGS & F_KS2) ? 512 : 256, (FLAGS & F_KS2) ? 1 : (DP == 96 ? ATTN96_BLOCKS : 2)) void attn_fwd_lds_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, DROP = FLAGS & F_DROP, DBITS = FLAGS & F_DBITS, KS2 = FLAGS & F_KS2;
  constexpr int NT = KS2 ? 512 : 256;
  // VROW: V is staged ROW-major straight from the projection output (like K) and the V^T fragments of the second product are
  // gathered with the LDS transpose read (ds_read_b64_tr_b16, two per 32 x 16 fragment): no transposed copy of V in HBM
  constexpr bool VROW = FLAGS & F_VROW;
  constexpr bool F16 = FLAGS & F_F16;
  constexpr int KROW = DP * 2, KCPR = DP / 8;                 // K row bytes, 16-B chunks per K row
  constexpr int K_BYTES = 64 * KROW, V_BYTES = DP * 128, STAGE = K_BYTES + V_BYTES;
  constexpr int NJK = 64 * KCPR / NT, NJV = DP * 8 / NT;      // DMA instructions per thread per stage (VROW: 64 rows x KCPR chunks = the same count)
  // everything lives in the dynamic region (a static array in front of it would shift its base off 16-B alignment)
  extern __shared__ __attribute__((aligned(16))) char sm[];  // 2 * STAGE bytes + 257-float bias LUT
  float* lut = reinterpret_cast<float*>(sm + 2 * STAGE);
  uint32_t* mbits = reinterpret_cast<uint32_t*>(sm + 2 * STAGE + 1040);   // [ATTN_MASK_WORDS] key-mask bitmap (MASK only)
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx_, h, b;
  attn_block(bx_, h, b);
  if (LUT) {
    for (int i = tid; i < 257; i += NT) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
  }
  const int wsub = KS2 ? (w >> 2) : 0;   // KS2: the 32-key half of every stage this wave takes
  const int q0 = (bx_ * 4 + (w & 3)) * 32;
  const bool active = q0 < p.Sq;  // wave-uniform; inactive waves still stage and hit the barriers
  const int q = q0 + l31;
  const bool q_ok = q < p.Sq;
  // (Measured in round 3 and not kept — the ViT's 257 = 2 * 128 + 1 = 4 * 64 + 1 tokens cost 87 us per block against 67 us at 256: (i) a
  // VALU path for the third query block's single row instead of staging 257 keys for it: 88.2 -> 86.6 us; (ii) the 257th KEY folded in
  // by a rank-1 update after four stages instead of a fifth stage: -1.6 us; (iii) no third block at all, the last full block appending
  // the row: 99.6 us, the serial tail sits on the head's critical path.  What costs is the third BLOCK per head — dispatch + start /
  // end latency at 216 VGPRs and 49 KB of LDS — not what it computes)
  bf16x8 qf[KS];
  load_rows<KS>(qf, p.Q.ptr + b * p.Q.bs + h * p.Q.hs, p.Q.rs, q, p.Sq, p.D, hi);
  float m_run = NEG_BIG, l_run = 0.f;
  f32x16 o[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(o[mt]);
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  if (MASK) attn_mask_bitmap(mbits, km, p.Skpad, tid, NT);   // (visible after the first stage barrier below)
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const uint32_t t_lane = DROP ? (row_id * (uint32_t)((p.Sk + 3) >> 2) + 2u * (uint32_t)hi) * MRB_H1 + mrb_lin_base(drop_seed, p.drop.site) : 0u;
  const float scale2 = p.scale * MRB_LOG2E;
  uint32_t* dbits_row = DBITS ? p.dbits + (long long)(b * p.H + h) * (p.Skpad >> 5) * p.Sqpad + q : nullptr;
  const uint32_t drop_tm1 = DROP ? (p.drop.thresh24 - 1u) * 0x10001u : 0u;   // (thresh - 1) in both 16-bit halves (attn_keep01)
  // the two far buckets of the relative-position bias (every key >= 128 positions before / after the query), as scalars
  const float bfar_lo = LUT ? attn_uniform(p.lut[h * 257] * MRB_LOG2E) : 0.f;
  const float bfar_hi = LUT ? attn_uniform(p.lut[h * 257 + 256] * MRB_LOG2E) : 0.f;

  // ---- staging: buffer resources span from this head's first element to the end of the tensor (reads past the head's rows stay
  // inside the tensor or return 0; whatever they return is finite and meets P == 0 / zero-padded Q)
  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vtbase = VROW ? p.V.ptr + b * p.V.bs + h * p.V.hs : p.Vt.ptr + b * p.Vt.bs + h * p.Vt.hs;
  const long long k_rem = ((long long)(p.B - 1 - b) * p.K.bs + (long long)(p.H - 1 - h) * p.K.hs + (long long)(p.Sk - 1) * p.K.rs + p.D) * 2;
  const long long v_rem = VROW ? ((long long)(p.B - 1 - b) * p.V.bs + (long long)(p.H - 1 - h) * p.V.hs + (long long)(p.Sk - 1) * p.V.rs + p.D) * 2
                               : ((long long)(p.B - 1 - b) * p.Vt.bs + (long long)(p.H - h) * p.Vt.hs) * 2;
  const uint32_t k_bytes = (uint32_t)(k_rem > 0xffffffffLL ? 0xffffffffLL : k_rem), v_bytes = (uint32_t)(v_rem > 0xffffffffLL ? 0xffffffffLL : v_rem);
  auto ksw = [](int row) { return DP == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };
  uint32_t vK[NJK], vV[NJV];
#pragma unroll
  for (int j = 0; j < NJK; ++j) {
    const int g = j * NT + tid, row = g / KCPR, c = g % KCPR;
    vK[j] = (uint32_t)((long long)row * p.K.rs * 2) + (uint32_t)((c ^ ksw(row)) * 16);
  }
#pragma unroll
  for (int j = 0; j < NJV; ++j) {
    const int g = j * NT + tid, d = g >> 3, c = g & 7;
    if (VROW) vV[j] = (uint32_t)((long long)(g / KCPR) * p.V.rs * 2) + (uint32_t)((g % KCPR) * 16);   // row-major image, no swizzle (see vtr)
    else vV[j] = (uint32_t)((long long)d * p.Vt.ds * 2) + (uint32_t)((c ^ ((d >> 1) & 7)) * 16);
  }
  auto stage = [&](int st, int buf) {
    attn_stage_dma<NJK, NJV, NT>(sm + buf * STAGE, sm + buf * STAGE + K_BYTES, kbase, k_bytes, vtbase, v_bytes, vK, vV, w, (uint32_t)((long long)st * 64 * p.K.rs * 2),
                             VROW ? (uint32_t)((long long)st * 64 * p.V.rs * 2) : (uint32_t)(st * 128));
  };
  // VROW fragment gather: each 16-lane group reads a [4 keys][16 d] block, lane i supplying the address of 4 contiguous d of key i / 4;
  // the hardware hands lane c the 4 keys of column c (tools/probes/tr_probe.hip checks exactly this map).  Lane (d = 32 mt + l31, hi)
  // needs keys 32 sub + 16 hf + 8 hi + 0..7: two reads 4 rows apart, the (sub, hf, mt) part is an immediate offset.  With 192-B rows
  // (head_dim 88 -> 96) the four rows of a block start 48 banks apart: conflict-free without a swizzle.
  const int vtr = K_BYTES + (8 * hi + ((lane & 15) >> 2)) * KROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  // fragment addresses inside a stage (bytes)
  const int krow = perm23(l31);
  // Offsets of sub-tile 0; sub-tile 1 (keys 32..63 of the stage) lies 32 K rows further (the row swizzle ignores bit 5 of the row) and
  // in the other 64-B half of the V^T rows (chunk index ^ 4).  ONE instance of the tile body serves both sub-tiles and the edge tiles
  // (round 3): with four unrolled instances (sub x edge) the compiler ping-ponged the O accumulators between two register sets and paid
  // 16 v_mov_b64 per tile in the common no-rescale path.
  int k_off[KS], v_off[MT][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) k_off[s] = krow * KROW + (((2 * s + hi) ^ ksw(krow)) << 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int d = mt * 32 + l31;
      v_off[mt][hf] = K_BYTES + d * 128 + (((2 * hf + hi) ^ ((d >> 1) & 7)) << 4);
    }

  auto tile = [&](const bool EDGE, int k0, const char* base, int sub, uint32_t vmask) {
    // VROW: the transpose reads of this tile's V fragments go out FIRST - their latency then lies under the QK^T MFMAs and the softmax
    // arithmetic instead of in front of the second product (issued where they are consumed the waves sat parked 40 % longer)
    typedef short v4s_t __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4s_t* tr_ptr_t;
    v4s_t vtr_r[VROW ? MT : 1][4];
#ifdef EXP_VROW_OLDREAD   // ablation (wrong results): row-major staging, but the old 16-B fragment reads
    constexpr bool VTR = false;
#else
    constexpr bool VTR = VROW;
#endif
    if (VTR) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const char* vb = base + vtr + sub * (32 * KROW) + mt * 64;
        vtr_r[mt][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(vb));
        vtr_r[mt][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(vb + 4 * KROW));
        vtr_r[mt][2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(vb + 16 * KROW));
        vtr_r[mt][3] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr_t)(vb + 20 * KROW));
      }
    }
    f32x16 sacc;
    zero16(sacc);
#pragma unroll
    for (int s = 0; s < KS; ++s)
      sacc = mfma32x16<F16>(*reinterpret_cast<const bf16x8*>(base + sub * (32 * KROW) + k_off[s]), qf[s], sacc);
    // lane (q, hi), register r  <->  key k0 + 16*(r>>3) + 8*hi + (r&7)
    // Round 5.  The common tile (every ViT / Q-Former tile; ~85 % of the T5 encoder's at S = 2012) has ONE bias bucket: the bias is a
    // per-head SCALAR held in an SGPR since before the loop (no LDS read + wait + readfirstlane per tile), the running maximum is taken on
    // the RAW accumulators (scale2 > 0: s -> s * scale2 + b is monotonic, so max and fma commute bit for bit) and every probability is ONE
    // fma + ONE exp: exp2(s * scale2 + (b - m)).  The general tile (per-element bias, masked keys) prepares finished scores and runs the
    // same exponent line with multiplier 1 (fma(s, 1, -m) == s - m exactly): one rescale site, one exponent site for both.
    float mx, emul, ebase;   // emul / ebase: wave-uniform
    const bool far_tile = !LUT || k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128;   // wave-uniform: one bias bucket for the whole tile
    if (!EDGE && far_tile) {
      emul = scale2;
      ebase = LUT ? (k0 > q0 ? bfar_hi : bfar_lo) : 0.f;
      float mraw = fmaxf(sacc[0], sacc[1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mraw = fmaxf(fmaxf(mraw, sacc[r]), sacc[r + 1]);
      mx = __builtin_fmaf(mraw, scale2, ebase);
    } else {   // (the finished scores replace the accumulators IN PLACE: both arms hand the same registers to the exponent line)
      emul = 1.0f;
      ebase = 0.f;
      if (LUT) {
        float bias[16];
        const int relbase = k0 + 8 * hi - q;
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = lut[max(-128, min(128, relbase + 16 * (r >> 3) + (r & 7))) + 128];
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = __builtin_fmaf(sacc[r], scale2, bias[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] *= scale2;
      }
      if (EDGE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
          const bool ok = key < p.Sk && ((vmask >> r) & 1u);
          sacc[r] = ok ? sacc[r] : NEG_INF;
        }
      }
      mx = fmaxf(sacc[0], sacc[1]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sacc[r]), sacc[r + 1]);
    }
    mx = attn_max_x32(mx);
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0) {  // lazy rescale
      const float m_new = fmaxf(m_run, mx);
      const float alpha = ex2(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
    }
    const float cadd = ebase - m_run;
    float psum = 0.f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = ex2(__builtin_fmaf(sacc[r], emul, cadd));
      psum += pv[r];
    }
    l_run += psum;
    // the probabilities are packed FIRST; the dropout mask is applied to the packed pairs (draws v3, see attn_keep01)
    union { bf16x8 v8[2]; uint32_t u[8]; } pk;
#pragma unroll
    for (int i = 0; i < 8; ++i) pk.u[i] = pack2x<F16>(pv[2 * i], pv[2 * i + 1]);
    if (DROP) {
      uint32_t bits = 0;  // keep bit of register pair i at positions i and 16 + i; shifted by 8 * hi it is the lane half's share of the tile word
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {  // cq = 2c + qd: registers 4 cq .. 4 cq + 3 = pairs 2 cq, 2 cq + 1
        const uint32_t hsh = mrb_lin_fin24(t_lane + (uint32_t)((k0 >> 2) + 4 * (cq >> 1) + (cq & 1)) * MRB_H1);
        const uint32_t ka = attn_keep01(hsh, drop_tm1), kb = attn_keep01(attn_rot8(hsh), drop_tm1);
        pk.u[2 * cq] = attn_pkmul(pk.u[2 * cq], ka);
        pk.u[2 * cq + 1] = attn_pkmul(pk.u[2 * cq + 1], kb);
        if (DBITS) bits |= (ka << (2 * cq)) | (kb << (2 * cq + 1));
      }
      if (DBITS) {
        const uint32_t wbits = attn_or_x32(bits << (8 * hi));
        if (hi == 0) dbits_row[(long long)(k0 >> 5) * p.Sqpad] = wbits;
      }
    }
    const bf16x8 pf0 = pk.v8[0], pf1 = pk.v8[1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (VTR) {
        const v4s_t a0 = vtr_r[mt][0], a1 = vtr_r[mt][1], b0 = vtr_r[mt][2], b1 = vtr_r[mt][3];
        const bf16x8 vf0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, vf1 = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        o[mt] = mfma32x16<F16>(vf0, pf0, o[mt]);
        o[mt] = mfma32x16<F16>(vf1, pf1, o[mt]);
      } else {
        o[mt] = mfma32x16<F16>(*reinterpret_cast<const bf16x8*>(base + (v_off[mt][0] ^ (sub << 6))), pf0, o[mt]);
        o[mt] = mfma32x16<F16>(*reinterpret_cast<const bf16x8*>(base + (v_off[mt][1] ^ (sub << 6))), pf1, o[mt]);
      }
    }
  };

  const int nst = (p.Sk + 63) >> 6, ntile = (p.Sk + 31) >> 5;
  stage(0, 0);
  uint32_t vm0 = 0xffffu, vm1 = 0xffffu;
  bool part0 = false, part1 = false;   // wave-uniform: the tile holds masked keys (a tile of valid keys only skips the per-score selects)
#pragma unroll 1
  for (int t = KS2 ? wsub : 0; t < (KS2 ? 2 * nst : ntile); t += KS2 ? 2 : 1) {  // one 32-key tile per trip (KS2: this wave's half of the stage)
    const int st = t >> 1, sub = t & 1;
    if (KS2 || sub == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // stage st has landed for every wave; every wave is done reading the other buffer
      if (MASK) {  // Skpad is a multiple of 32: the second half of the last stage may lie past the row
        vm0 = mask_bits_lds(mbits, 2 * st, hi);
        vm1 = st * 64 + 32 < p.Skpad ? mask_bits_lds(mbits, 2 * st + 1, hi) : 0u;
        part0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mbits[2 * st]) != 0xffffffffu;
        part1 = st * 64 + 32 < p.Skpad ? (uint32_t)__builtin_amdgcn_readfirstlane((int)mbits[2 * st + 1]) != 0xffffffffu : true;
      }
      if (st + 1 < nst) stage(st + 1, (st + 1) & 1);
    }
    if (active && 32 * t < p.Sk) tile((MASK && (sub ? part1 : part0)) || 32 * t + 32 > p.Sk, 32 * t, sm + (st & 1) * STAGE, sub, sub ? vm1 : vm0);
  }
  if (KS2) {  // merge the two key halves of every query tile: wave w + 4 hands (m, l, O) to wave w through the (now idle) stage buffers
    float* mg = reinterpret_cast<float*>(sm) + (w & 3) * ((MT * 16 + 2) * 64);
    __syncthreads();
    if (wsub == 1) {
      mg[lane] = m_run;
      mg[64 + lane] = l_run;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mg[(2 + mt * 16 + r) * 64 + lane] = o[mt][r];
    }
    __syncthreads();
    if (wsub == 1) return;
    const float m1 = mg[lane], m_all = fmaxf(m_run, m1);
    const float f0 = ex2(m_run - m_all), f1 = ex2(m1 - m_all);
    l_run = l_run * f0 + mg[64 + lane] * f1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mt][r] = o[mt][r] * f0 + mg[(2 + mt * 16 + r) * 64 + lane] * f1;
    m_run = m_all;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = l_tot > 0.f ? (DROP ? p.drop.inv_keep : 1.0f) / l_tot : 0.f;
  if (q_ok) {
    bf16_t* op = const_cast<bf16_t*>(p.O.ptr) + b * p.O.bs + h * p.O.hs + (long long)q * p.O.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D)
          *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2x<F16>(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv),
                                                          pack2x<F16>(o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv));
      }
    if (p.LSE && hi == 0) p.LSE[((long long)(b * p.H + h)) * p.Sqpad + q] = m_run * MRB_LN2 + __logf(fmaxf(l_tot, 1e-37f));
  }
}

// ---- backward, part 1: dQ (and Delta = rowsum(dO * O), needed by part 2).  Same ownership as the forward.
template <int DP, int FLAGS, int NSW = 4>
__global__ __launch_bounds__(NSW * 64, NSW == 4 ? 2 : 1) void attn_bwd_dq_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP, SPLIT = FLAGS & F_SPLIT;
  constexpr bool XS = FLAGS & F_XS;
  static_assert(NSW == 4 || SPLIT, "more than four waves only in the key-split form");
  static_assert(!XS || (SPLIT && !CAUSAL), "the cross-block key split extends the key-split form");
  __shared__ float lut[LUT ? 257 : 1];
  __shared__ float red[SPLIT ? (NSW - 1) * MT * 16 * 64 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += NSW * 64) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
    __syncthreads();
  }
  const int q0 = XS ? 0 : SPLIT ? blockIdx.x * 32 : (blockIdx.x * 4 + w) * 32;
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
  const float lse2 = q_ok ? p.LSE[stat_off] * MRB_LOG2E : 0.f;
  if (q_ok && hi == 0 && (!SPLIT || w == 0) && (!XS || blockIdx.x == 0)) p.Delta[stat_off] = delta;
  const float scale2 = p.scale * MRB_LOG2E;
  const float keep_scale = DROP ? p.drop.inv_keep : 1.0f;

  f32x16 dq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(dq[mt]);
  const int xs_tiles = XS ? ((p.Sk + 31) / 32 + (int)gridDim.x - 1) / (int)gridDim.x : 0;   // XS: this block's chunk of whole key tiles
  const int kend = XS ? min(p.Sk, ((int)blockIdx.x + 1) * xs_tiles * 32) : CAUSAL ? min(p.Sk, q0 + 32) : p.Sk;
  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vbase = p.V.ptr + b * p.V.bs + h * p.V.hs;
  const bf16_t* ktbase = p.Kt.ptr + b * p.Kt.bs + h * p.Kt.hs;
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const uint32_t t_lane = DROP ? (row_id * (uint32_t)((p.Sk + 3) >> 2) + 2u * (uint32_t)hi) * MRB_H1 + mrb_lin_base(drop_seed, p.drop.site) : 0u;
  const int kstart = (XS ? (int)blockIdx.x * xs_tiles * 32 : 0) + (SPLIT ? w * 32 : 0), kstep = SPLIT ? NSW * 32 : 32;

  bf16x8 kcur[KS], vcur[KS];
  load_rows<KS>(kcur, kbase, p.K.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  load_rows<KS>(vcur, vbase, p.V.rs, kstart + perm23(l31), p.Sk, p.D, hi);
  auto tile = [&](auto edge_c, int k0) {
    constexpr bool EDGE = decltype(edge_c)::value;
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
    tile_scores<LUT, 1>(sv, sacc, scale2, lut, k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128, k0 > q0 ? 256 : 0, k0 + 8 * hi - q);
    float kf[16];  // dropout factor of dP: keep ? 1/(1-p) : 0
#pragma unroll
    for (int r = 0; r < 16; ++r) kf[r] = keep_scale;
    if (DROP) {
      attn_keep16(t_lane, k0, p.drop.thresh24, [&](int r, bool kp) { kf[r] = kp ? keep_scale : 0.f; });
    }
    float ds[16];  // dS / scale (the scale is applied once to dQ at the end)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pr = ex2(sv[r] - lse2);
      if (EDGE) {
        const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
        bool ok = key < p.Sk && ((vmask >> r) & 1u);
        if (CAUSAL) ok = ok && (key <= q);
        pr = ok ? pr : 0.f;
      }
      ds[r] = pr * (kf[r] * dpacc[r] - delta);
    }
    const bf16x8 f0 = pack8(ds), f1 = pack8(ds + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[mt][0], f0, dq[mt], 0, 0, 0);
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[mt][1], f1, dq[mt], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { kcur[s] = knext[s]; vcur[s] = vnext[s]; }
  };
  for (int k0 = kstart; k0 < kend; k0 += kstep) {
    const bool edge = MASK || k0 + 32 > p.Sk || (CAUSAL && k0 + 31 > q0);
    if (edge) tile(BoolC<true>(), k0);
    else tile(BoolC<false>(), k0);
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
    for (int j = 0; j < NSW - 1; ++j)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[mt][i] += red[(j * MT * 16 + mt * 16 + i) * 64 + lane];
  }
  if constexpr (XS) {   // (wave 0 only from here) publish this chunk's partial dQ; the last arriver adds all chunks in chunk order
    constexpr int NV4 = MT * 4;
    const int nx = (int)gridDim.x;
    float* base = p.xs_ws + (long long)(b * p.H + h) * nx * (NV4 * 64 * 4);
    f32x4 v[NV4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) v[mt * 4 + g] = f32x4{dq[mt][4 * g], dq[mt][4 * g + 1], dq[mt][4 * g + 2], dq[mt][4 * g + 3]};
    xs_publish<NV4>(base + (long long)blockIdx.x * (NV4 * 64 * 4), v, lane);
    if (!xs_last(p.xs_cnt + b * p.H + h, nx, lane)) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) zero16(dq[mt]);
    for (int j = 0; j < nx; ++j) {
      xs_fetch<NV4>(base + (long long)j * (NV4 * 64 * 4), v, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          dq[mt][4 * g] += v[mt * 4 + g][0]; dq[mt][4 * g + 1] += v[mt * 4 + g][1]; dq[mt][4 * g + 2] += v[mt * 4 + g][2]; dq[mt][4 * g + 3] += v[mt * 4 + g][3];
        }
    }
  }
  if (q_ok) {
    bf16_t* op = const_cast<bf16_t*>(p.dQ.ptr) + b * p.dQ.bs + h * p.dQ.hs + (long long)q * p.dQ.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D)
          *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(dq[mt][4 * g] * p.scale, dq[mt][4 * g + 1] * p.scale),
                                                          pack2bf(dq[mt][4 * g + 2] * p.scale, dq[mt][4 * g + 3] * p.scale));
      }
  }
}

// ---- dQ with the key-side tiles (K rows, V rows, K^T) shared through LDS: same staging scheme as attn_fwd_lds_kernel (head_dim 64).
template <int FLAGS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_lds_kernel(const AttnArgs p) {
  constexpr int KS = 4, MT = 2;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, DROP = FLAGS & F_DROP, DBITS = FLAGS & F_DBITS;
  constexpr int T_BYTES = 64 * 128, STAGE = 3 * T_BYTES;  // K rows | V rows | K^T, 64 keys each
  // everything lives in the dynamic region (a static array in front of it would shift its base off 16-B alignment)
  extern __shared__ __attribute__((aligned(16))) char sm[];  // 2 * STAGE bytes + 257-float bias LUT
  float* lut = reinterpret_cast<float*>(sm + 2 * STAGE);
  uint32_t* mbits = reinterpret_cast<uint32_t*>(sm + 2 * STAGE + 1040);   // [ATTN_MASK_WORDS] key-mask bitmap (MASK only)
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx_, h, b;
  attn_block(bx_, h, b);
  if (LUT) {
    for (int i = tid; i < 257; i += 256) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
  }
  const int q0 = (bx_ * 4 + w) * 32;
  const bool active = q0 < p.Sq;
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
  const float lse2 = q_ok ? p.LSE[stat_off] * MRB_LOG2E : 0.f;
  if (q_ok && hi == 0) p.Delta[stat_off] = delta;
  const float scale2 = p.scale * MRB_LOG2E;
  const float keep_scale = DROP ? p.drop.inv_keep : 1.0f;
  const float bfar_lo = LUT ? attn_uniform(p.lut[h * 257] * MRB_LOG2E) : 0.f;        // the two far buckets of the bias, as scalars
  const float bfar_hi = LUT ? attn_uniform(p.lut[h * 257 + 256] * MRB_LOG2E) : 0.f;

  f32x16 dq[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) zero16(dq[mt]);
  const int* km = MASK ? p.kmask + (long long)b * p.Skpad : nullptr;
  if (MASK) attn_mask_bitmap(mbits, km, p.Skpad, tid, 256);
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t row_id = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq + (uint32_t)q;
  const uint32_t t_lane = DROP ? (row_id * (uint32_t)((p.Sk + 3) >> 2) + 2u * (uint32_t)hi) * MRB_H1 + mrb_lin_base(drop_seed, p.drop.site) : 0u;

  const bf16_t* kbase = p.K.ptr + b * p.K.bs + h * p.K.hs;
  const bf16_t* vbase = p.V.ptr + b * p.V.bs + h * p.V.hs;
  const bf16_t* ktbase = p.Kt.ptr + b * p.Kt.bs + h * p.Kt.hs;
  auto clamp32 = [](long long v) { return (uint32_t)(v > 0xffffffffLL ? 0xffffffffLL : v); };
  const uint32_t k_bytes = clamp32(((long long)(p.B - 1 - b) * p.K.bs + (long long)(p.H - 1 - h) * p.K.hs + (long long)(p.Sk - 1) * p.K.rs + p.D) * 2);
  const uint32_t v_bytes = clamp32(((long long)(p.B - 1 - b) * p.V.bs + (long long)(p.H - 1 - h) * p.V.hs + (long long)(p.Sk - 1) * p.V.rs + p.D) * 2);
  const uint32_t kt_bytes = clamp32(((long long)(p.B - 1 - b) * p.Kt.bs + (long long)(p.H - h) * p.Kt.hs) * 2);
  uint32_t vK[2], vV[2], vT[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = j * 256 + tid, row = g >> 3, c = g & 7, sw = (c ^ ((row >> 1) & 7)) * 16;
    vK[j] = (uint32_t)((long long)row * p.K.rs * 2) + (uint32_t)sw;
    vV[j] = (uint32_t)((long long)row * p.V.rs * 2) + (uint32_t)sw;
    vT[j] = (uint32_t)((long long)row * p.Kt.ds * 2) + (uint32_t)sw;
  }
  auto stage = [&](int st, int buf) {
    char* base = sm + buf * STAGE;
    attn_stage_dma<2, 2>(base, base + T_BYTES, kbase, k_bytes, vbase, v_bytes, vK, vV, w, (uint32_t)((long long)st * 64 * p.K.rs * 2),
                         (uint32_t)((long long)st * 64 * p.V.rs * 2));
    attn_stage_dma<2, 0>(base + 2 * T_BYTES, base, ktbase, kt_bytes, ktbase, kt_bytes, vT, vT, w, (uint32_t)(st * 128), 0u);
  };

  const int krow = perm23(l31);
  // offsets of sub-tile 0; sub-tile 1: rows 32 further (+ 4096 B, the swizzle ignores bit 5), K^T chunk index ^ 4 (^ 64 B).  ONE
  // instance of the tile body for both sub-tiles and the edge tiles (see attn_fwd_lds_kernel).
  int k_off[KS], t_off[MT][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) k_off[s] = krow * 128 + (((2 * s + hi) ^ ((krow >> 1) & 7)) << 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int d = mt * 32 + l31;
      t_off[mt][hf] = 2 * T_BYTES + d * 128 + (((2 * hf + hi) ^ ((d >> 1) & 7)) << 4);
    }

  // EDGE is a COMPILE-TIME flag of the tile body (round 5): as a run-time argument of the inlined lambda the compiler if-converted the
  // edge handling — every interior tile computed 16 x (key index, range compare, mask-bit test, two selects) and threw them away
  // (~100 of ~280 VALU instructions per tile, found in the ISA); the call site now branches (wave-uniform) between two instances.
  auto tile = [&](auto edge_c, int k0, const char* base, int sub, uint32_t vmask, uint32_t dword) {
    constexpr bool EDGE = decltype(edge_c)::value;
    f32x16 sacc, dpacc;
    zero16(sacc);
    zero16(dpacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + sub * 4096 + k_off[s]), qf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + T_BYTES + sub * 4096 + k_off[s]), dof[s], dpacc, 0, 0, 0);
    }
    // P = exp2(s * emul + cadd): the one-bucket tile folds scale, bias and -LSE into ONE fma per score (see attn_fwd_lds_kernel); the
    // general tile finishes its scores in place first
    float emul, cadd;
    const bool far_tile = !LUT || k0 - (q0 + 31) >= 128 || (q0 - (k0 + 31)) >= 128;   // wave-uniform
    if (!EDGE && far_tile) {
      emul = scale2;
      cadd = (LUT ? (k0 > q0 ? bfar_hi : bfar_lo) : 0.f) - lse2;
    } else {
      emul = 1.0f;
      cadd = -lse2;
      if (LUT) {
        float bias[16];
        const int relbase = k0 + 8 * hi - q;
#pragma unroll
        for (int r = 0; r < 16; ++r) bias[r] = lut[max(-128, min(128, relbase + 16 * (r >> 3) + (r & 7))) + 128];
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = __builtin_fmaf(sacc[r], scale2, bias[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] *= scale2;
      }
      if (EDGE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);
          const bool ok = key < p.Sk && ((vmask >> r) & 1u);
          sacc[r] = ok ? sacc[r] : NEG_INF;     // exp2(-inf + finite) = 0
        }
      }
    }
    float kf[16];  // dropout factor of dP: keep ? 1/(1-p) : 0
#pragma unroll
    for (int r = 0; r < 16; ++r) kf[r] = keep_scale;
    if (DROP && DBITS) {  // keep bits stored by the forward (layout: attn_bitpos_reg)
      const int wsh = (int)(dword >> (8 * hi));
#pragma unroll
      for (int r = 0; r < 16; ++r)   // v_bfe_i32: the key's bit as 0 / -1, then one AND (instead of and + compare + select)
        kf[r] = __builtin_bit_cast(float, __builtin_amdgcn_sbfe(wsh, attn_bitpos_reg(r), 1) & __builtin_bit_cast(int, keep_scale));
    } else if (DROP) {
      attn_keep16(t_lane, k0, p.drop.thresh24, [&](int r, bool kp) { kf[r] = kp ? keep_scale : 0.f; });
    }
    float ds[16];  // dS / scale (the scale is applied once to dQ at the end)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = ex2(__builtin_fmaf(sacc[r], emul, cadd));
      ds[r] = pr * (kf[r] * dpacc[r] - delta);
    }
    const bf16x8 f0 = pack8(ds), f1 = pack8(ds + 8);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + (t_off[mt][0] ^ (sub << 6))), f0, dq[mt], 0, 0, 0);
      dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + (t_off[mt][1] ^ (sub << 6))), f1, dq[mt], 0, 0, 0);
    }
  };

  const int nst = (p.Sk + 63) >> 6;
  const uint32_t* dbits_row = DBITS ? p.dbits + (long long)(b * p.H + h) * (p.Skpad >> 5) * p.Sqpad + min(q, p.Sqpad - 1) : nullptr;
  const int nkt = p.Skpad >> 5;
  uint32_t dw0 = 0, dw1 = 0;  // keep-bit words of the NEXT stage's two sub-tiles (loaded one stage ahead)
  stage(0, 0);
  if (DBITS) {
    dw0 = dbits_row[0];
    dw1 = 1 < nkt ? dbits_row[p.Sqpad] : 0u;
  }
  for (int st = 0; st < nst; ++st) {
    uint32_t vm0 = 0xffffu, vm1 = 0xffffu;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool part0 = false, part1 = false;   // wave-uniform: the tile holds masked keys (see the forward)
    if (MASK) {
      vm0 = mask_bits_lds(mbits, 2 * st, hi);
      vm1 = st * 64 + 32 < p.Skpad ? mask_bits_lds(mbits, 2 * st + 1, hi) : 0u;
      part0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mbits[2 * st]) != 0xffffffffu;
      part1 = st * 64 + 32 < p.Skpad ? (uint32_t)__builtin_amdgcn_readfirstlane((int)mbits[2 * st + 1]) != 0xffffffffu : true;
    }
    const uint32_t cw0 = dw0, cw1 = dw1;
    if (st + 1 < nst) {
      stage(st + 1, (st + 1) & 1);
      if (DBITS) {
        dw0 = dbits_row[(long long)(2 * st + 2) * p.Sqpad];
        dw1 = 2 * st + 3 < nkt ? dbits_row[(long long)(2 * st + 3) * p.Sqpad] : 0u;
      }
    }
    const char* base = sm + (st & 1) * STAGE;
    if (active) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {   // (sub is a compile-time constant here: two instances, the edge test stays a run-time branch)
        const int k0 = st * 64 + 32 * sub;
        if (k0 < p.Sk) {
          if ((MASK && (sub ? part1 : part0)) || k0 + 32 > p.Sk) tile(BoolC<true>{}, k0, base, sub, sub ? vm1 : vm0, sub ? cw1 : cw0);
          else tile(BoolC<false>{}, k0, base, sub, sub ? vm1 : vm0, sub ? cw1 : cw0);
        }
      }
    }
  }
  if (q_ok) {
    bf16_t* op = const_cast<bf16_t*>(p.dQ.ptr) + b * p.dQ.bs + h * p.dQ.hs + (long long)q * p.dQ.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D)
          *reinterpret_cast<uint2*>(op + d0) = make_uint2(pack2bf(dq[mt][4 * g] * p.scale, dq[mt][4 * g + 1] * p.scale),
                                                          pack2bf(dq[mt][4 * g + 2] * p.scale, dq[mt][4 * g + 3] * p.scale));
      }
  }
}

// ---- backward, part 2: dK, dV.  One wave owns 32 keys and walks the query tiles.
// Round 5: the kernel is compiled for TWO resident blocks per CU.  Left to itself the compiler took 398 registers (256 + 142 accumulation
// registers: two instances of the tile body and the next tile's prefetched fragments) — ONE wave per SIMD, so the Q-Former's cross-attention
// backward (32 queries x 257 keys: one tile per wave, a chain of dependent loads) ran 2160 blocks as 8.4 rounds of exposed latency:
// 142 us per launch, six launches per step.
#ifndef ATTN_DKV_BLOCKS
#define ATTN_DKV_BLOCKS 2
#endif
template <int DP, int FLAGS>
__global__ __launch_bounds__(256, ATTN_DKV_BLOCKS) void attn_bwd_dkv_kernel(const AttnArgs p) {
  constexpr int KS = DP / 16, MT = DP / 32;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP;
  __shared__ float lut[LUT ? 257 : 1];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, hi = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.z, h = blockIdx.y;
  if (LUT) {
    for (int i = threadIdx.x; i < 257; i += 256) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
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
  const float scale2 = p.scale * MRB_LOG2E;
  const float keep_scale = DROP ? p.drop.inv_keep : 1.0f;

  // ONE instance of the tile body, EDGE a run-time wave-uniform flag behind a real branch, operands fetched where they are used (round 5:
  // two compile-time instances + the next tile's prefetched fragments cost 398 registers, see above; every caller of this form has at most
  // 32 queries, i.e. ONE tile per wave — nothing to prefetch for)
  auto tile = [&](const bool EDGE, int q0) {
    bf16x8 qcur[KS], docur[KS];
    load_rows<KS>(qcur, qbase, p.Q.rs, q0 + perm23(l31), p.Sq, p.D, hi);
    load_rows<KS>(docur, dobase, p.dO.rs, q0 + perm23(l31), p.Sq, p.D, hi);
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
    tile_scores<LUT, -1>(sv, sacc, scale2, lut, kb0 - (q0 + 31) >= 128 || (q0 - (kb0 + 31)) >= 128, kb0 > q0 ? 256 : 0, key - q0 - 8 * hi);
    // two independent halves (8 query rows each): short live ranges for the per-element temporaries
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      bf16x8 dotf[MT], qtf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dotf[mt] = ld8(dotbase + (long long)(mt * 32 + l31) * p.dOt.ds + q0 + 8 * hi + 16 * c);
        qtf[mt] = ld8(qtbase + (long long)(mt * 32 + l31) * p.Qt.ds + q0 + 8 * hi + 16 * c);
      }
      const int qq0 = q0 + 16 * c + 8 * hi;  // Sqpad is a multiple of 32 -> in-bounds
      const float4 a0 = *reinterpret_cast<const float4*>(lsebase + qq0), a1 = *reinterpret_cast<const float4*>(lsebase + qq0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(delbase + qq0), b1 = *reinterpret_cast<const float4*>(delbase + qq0 + 4);
      const float lse[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, del[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      uint32_t draw[8];
      if (DROP) drop_draws8_keyowner(draw, bh_idx + (uint32_t)qq0, key, (p.Sk + 3) >> 2, mrb_lin_base(drop_seed, p.drop.site));
      float prs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) prs[j] = ex2(fmaf(lse[j], -MRB_LOG2E, sv[8 * c + j]));
      if (EDGE) {
        asm volatile("; edge tile" ::: "memory");   // (keeps this block behind a real branch: see attn_bwd_dkv_lds_kernel)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int qq = qq0 + j;
          bool ok = qq < p.Sq;
          if (CAUSAL) ok = ok && (key <= qq);
          prs[j] = ok ? prs[j] : 0.f;
        }
      }
      float pd[8], ds[8];  // pd: dropped P (without 1/(1-p));  ds: dS / scale  — both factors are applied once at the end
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * c + j;
        const float pr = prs[j];
        float kfr = keep_scale, prd = pr;
        if (DROP) {
          const bool keep = draw[j] >= p.drop.thresh24;
          kfr = keep ? keep_scale : 0.f;
          prd = keep ? pr : 0.f;
        }
        pd[j] = prd;
        ds[j] = pr * (kfr * dpacc[r] - del[j]);
      }
      const bf16x8 pc = pack8(pd), sc = pack8(ds);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf[mt], pc, dv[mt], 0, 0, 0);
        dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf[mt], sc, dk[mt], 0, 0, 0);
      }
    }
  };
  for (int q0 = qstart; q0 < p.Sq; q0 += 32) tile(q0 + 32 > p.Sq || (CAUSAL && q0 < kb0 + 31), q0);
  const float fk = key_ok ? p.scale : 0.f, fv = key_ok ? keep_scale : 0.f;  // masked keys: exact zeros
  if (key < p.Sk) {
    bf16_t* kp = const_cast<bf16_t*>(p.dK.ptr) + b * p.dK.bs + h * p.dK.hs + (long long)key * p.dK.rs;
    bf16_t* vp = const_cast<bf16_t*>(p.dV.ptr) + b * p.dV.bs + h * p.dV.hs + (long long)key * p.dV.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D) {
          *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dk[mt][4 * g] * fk, dk[mt][4 * g + 1] * fk), pack2bf(dk[mt][4 * g + 2] * fk, dk[mt][4 * g + 3] * fk));
          *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dv[mt][4 * g] * fv, dv[mt][4 * g + 1] * fv), pack2bf(dv[mt][4 * g + 2] * fv, dv[mt][4 * g + 3] * fv));
        }
      }
  }
}

// ---- dK/dV with the query-side tiles shared through LDS (head_dim 64).  The four key-waves of a block consume the SAME Q, dO,
// Q^T, dO^T, LSE, Delta tiles, so the block fetches each 64-query stage once with LDS-DMA (two stages, one barrier per stage, the
// DMA of stage t+1 flies while stage t is consumed) instead of four times from L2; MFMA operands are ds_read_b128 from rows that
// are XOR-swizzled at 16-B granularity through the DMA source address.
__device__ __forceinline__ void attn_stage_stats(char* dst, const float* lse, const float* del, uint32_t bytes, int lane, uint32_t soff) {
  // 64 LSE + 64 Delta floats: lanes 0..15 of one wave, 16 B each
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse), 0, (int)bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(del), 0, (int)bytes, 0x00020000);
  if (lane < 16) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (attn_lds_ptr_t)dst, 16, (uint32_t)lane * 16u, soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (attn_lds_ptr_t)(dst + 256), 16, (uint32_t)lane * 16u, soff, 0, 0);
  }
}

__device__ __forceinline__ void attn_stage_bits(char* dst, const uint32_t* src, uint32_t bytes, int lane, uint32_t soff) {
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, (int)bytes, 0x00020000);
  if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (attn_lds_ptr_t)dst, 16, (uint32_t)lane * 16u, soff, 0, 0);
}

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_lds_kernel(const AttnArgs p) {
  constexpr int KS = 4, MT = 2;
  constexpr bool LUT = FLAGS & F_LUT, MASK = FLAGS & F_MASK, CAUSAL = FLAGS & F_CAUSAL, DROP = FLAGS & F_DROP, DBITS = FLAGS & F_DBITS;
  constexpr int T_BYTES = 64 * 128, STAGE = 4 * T_BYTES + 512 + 1024;  // Q rows | dO rows | Q^T | dO^T | lse[64] delta[64] | keep bits 4 x [64]
  // everything lives in the dynamic region (a static array in front of it would shift its base off 16-B alignment)
  extern __shared__ __attribute__((aligned(16))) char sm[];  // 2 * STAGE bytes + 257-float bias LUT
  float* lut = reinterpret_cast<float*>(sm + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63, hi = lane >> 5, l31 = lane & 31;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bx_, h, b;
  attn_block(bx_, h, b);
  if (LUT) {
    for (int i = tid; i < 257; i += 256) lut[i] = p.lut[h * 257 + i] * MRB_LOG2E;
  }
  const int kb0 = (bx_ * 4 + w) * 32;
  const bool active = kb0 < p.Sk;
  const int key = kb0 + l31;
  bool key_ok = key < p.Sk;
  bf16x8 kf[KS], vf[KS];
  load_rows<KS>(kf, p.K.ptr + b * p.K.bs + h * p.K.hs, p.K.rs, key, p.Sk, p.D, hi);
  load_rows<KS>(vf, p.V.ptr + b * p.V.bs + h * p.V.hs, p.V.rs, key, p.Sk, p.D, hi);
  if (MASK) key_ok = key_ok && (p.kmask[(long long)b * p.Skpad + min(key, p.Skpad - 1)] != 0);
  f32x16 dk[MT], dv[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) { zero16(dk[mt]); zero16(dv[mt]); }
  const uint32_t drop_seed = DROP ? *p.drop.seed_ptr : 0u;
  const uint32_t bh_idx = (uint32_t)(b * p.H + h) * (uint32_t)p.Sq;
  const int st0 = CAUSAL ? bx_ * 2 : 0;  // first 64-query stage (block-uniform); the per-wave causal limit is applied below
  const float scale2 = p.scale * MRB_LOG2E;
  const float keep_scale = DROP ? p.drop.inv_keep : 1.0f;
  const float bfar_lo = LUT ? attn_uniform(p.lut[h * 257] * MRB_LOG2E) : 0.f;        // the two far buckets of the bias, as scalars
  const float bfar_hi = LUT ? attn_uniform(p.lut[h * 257 + 256] * MRB_LOG2E) : 0.f;

  const bf16_t* qbase = p.Q.ptr + b * p.Q.bs + h * p.Q.hs;
  const bf16_t* dobase = p.dO.ptr + b * p.dO.bs + h * p.dO.hs;
  const bf16_t* qtbase = p.Qt.ptr + b * p.Qt.bs + h * p.Qt.hs;
  const bf16_t* dotbase = p.dOt.ptr + b * p.dOt.bs + h * p.dOt.hs;
  const float* lsebase = p.LSE + ((long long)(b * p.H + h)) * p.Sqpad;
  const float* delbase = p.Delta + ((long long)(b * p.H + h)) * p.Sqpad;
  auto clamp32 = [](long long v) { return (uint32_t)(v > 0xffffffffLL ? 0xffffffffLL : v); };
  const uint32_t q_bytes = clamp32(((long long)(p.B - 1 - b) * p.Q.bs + (long long)(p.H - 1 - h) * p.Q.hs + (long long)(p.Sq - 1) * p.Q.rs + p.D) * 2);
  const uint32_t do_bytes = clamp32(((long long)(p.B - 1 - b) * p.dO.bs + (long long)(p.H - 1 - h) * p.dO.hs + (long long)(p.Sq - 1) * p.dO.rs + p.D) * 2);
  const uint32_t t_bytes = clamp32(((long long)(p.B - 1 - b) * p.Qt.bs + (long long)(p.H - h) * p.Qt.hs) * 2);
  const uint32_t st_bytes = clamp32((long long)(p.B * p.H - (b * p.H + h)) * p.Sqpad * 4);
  // keep bits of this wave's key tile: row (bh, kt = kb0 / 32) of the [B*H, Skpad/32, Sqpad] word array
  const long long bits_row_idx = (long long)(b * p.H + h) * (p.Skpad >> 5) + (kb0 >> 5);
  const uint32_t* bits_row = DBITS ? p.dbits + bits_row_idx * p.Sqpad : nullptr;
  const uint32_t bits_bytes = DBITS ? clamp32(((long long)p.B * p.H * (p.Skpad >> 5) - bits_row_idx) * p.Sqpad * 4) : 0u;
  uint32_t vQ[2], vDO[2], vT[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int g = j * 256 + tid, row = g >> 3, c = g & 7, sw = (c ^ ((row >> 1) & 7)) * 16;
    vQ[j] = (uint32_t)((long long)row * p.Q.rs * 2) + (uint32_t)sw;
    vDO[j] = (uint32_t)((long long)row * p.dO.rs * 2) + (uint32_t)sw;
    vT[j] = (uint32_t)((long long)row * p.Qt.ds * 2) + (uint32_t)sw;  // Qt and dOt share their geometry
  }
  auto stage = [&](int st, int buf) {
    char* base = sm + buf * STAGE;
    attn_stage_dma<2, 2>(base, base + T_BYTES, qbase, q_bytes, dobase, do_bytes, vQ, vDO, w, (uint32_t)((long long)st * 64 * p.Q.rs * 2),
                         (uint32_t)((long long)st * 64 * p.dO.rs * 2));
    attn_stage_dma<2, 2>(base + 2 * T_BYTES, base + 3 * T_BYTES, qtbase, t_bytes, dotbase, t_bytes, vT, vT, w, (uint32_t)(st * 128), (uint32_t)(st * 128));
    if (w == 0) attn_stage_stats(base + 4 * T_BYTES, lsebase, delbase, st_bytes, lane, (uint32_t)(st * 256));
    if (DBITS && active) attn_stage_bits(base + 4 * T_BYTES + 512 + w * 256, bits_row, bits_bytes, lane, (uint32_t)(st * 256));
  };

  const int frow = perm23(l31);  // fragment row of the row tiles (the MFMA row permutation)
  const int kbitpos = attn_bitpos_key(l31);   // where the forward put this lane's key in a keep-bit word (attn_bitpos_key)
  int r_off[KS], t_off[MT][2];  // rows of sub-tile 1 sit 32 * 128 B further (the swizzle (row >> 1) & 7 ignores bit 5); its Q^T / dO^T
  // chunks have index ^ 4 (^ 64 B).  ONE instance of the tile body for both sub-tiles and the edge tiles (see attn_fwd_lds_kernel).
#pragma unroll
  for (int s = 0; s < KS; ++s) r_off[s] = frow * 128 + (((2 * s + hi) ^ ((frow >> 1) & 7)) << 4);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int d = mt * 32 + l31;
      t_off[mt][hf] = 2 * T_BYTES + d * 128 + (((2 * hf + hi) ^ ((d >> 1) & 7)) << 4);
    }

  // EDGE is a run-time, wave-uniform flag; its block holds an `asm volatile` so that it stays a BRANCH (round 5): left to itself the
  // compiler if-converted the edge handling and every interior tile computed 16 x (query index, range compare, two selects) for nothing
  // (64 VALU instructions per tile, found in the ISA).  (Two compile-time instances, as in the dQ kernel, cost 256 VGPRs + scratch here.)
  auto tile = [&](const bool EDGE, int q0, const char* base, int sub) {
    f32x16 sacc, dpacc;
    zero16(sacc);
    zero16(dpacc);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + sub * 4096 + r_off[s]), kf[s], sacc, 0, 0, 0);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + T_BYTES + sub * 4096 + r_off[s]), vf[s], dpacc, 0, 0, 0);
    }
    // lane (key, hi), register r  <->  query q0 + 16*(r>>3) + 8*hi + (r&7)
    float sv[16];
    if (LUT && (kb0 - (q0 + 31) >= 128 || (q0 - (kb0 + 31)) >= 128)) {   // wave-uniform: one bias bucket, a scalar held since before the loop
      const float bconst = kb0 > q0 ? bfar_hi : bfar_lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = __builtin_fmaf(sacc[r], scale2, bconst);
    } else {
      tile_scores<LUT, -1>(sv, sacc, scale2, lut, false, 0, key - q0 - 8 * hi);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two independent halves (8 query rows each): short live ranges
      const float* sp = reinterpret_cast<const float*>(base + 4 * T_BYTES) + 32 * sub + 16 * c + 8 * hi;
      const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(sp + 64), b1 = *reinterpret_cast<const float4*>(sp + 68);
      const float lse[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, del[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      int keepm[8];   // 0 / -1 per query row: the dropped values are cleared with one AND each
      if (DROP && DBITS) {  // stored keep bits: word j = query row q0 + 16c + 8hi + j, bit kbitpos = this lane's key
        const uint32_t* bp = reinterpret_cast<const uint32_t*>(base + 4 * T_BYTES + 512 + w * 256) + 32 * sub + 16 * c + 8 * hi;
        const uint4 w0 = *reinterpret_cast<const uint4*>(bp), w1 = *reinterpret_cast<const uint4*>(bp + 4);
        const uint32_t ws[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) keepm[j] = __builtin_amdgcn_sbfe((int)ws[j], kbitpos, 1);
      } else if (DROP) {
        uint32_t draw[8];
        drop_draws8_keyowner(draw, bh_idx + (uint32_t)(q0 + 16 * c + 8 * hi), key, (p.Sk + 3) >> 2, mrb_lin_base(drop_seed, p.drop.site));
#pragma unroll
        for (int j = 0; j < 8; ++j) keepm[j] = draw[j] >= p.drop.thresh24 ? -1 : 0;
      }
      float pd[8], ds[8];  // pd: dropped P (without 1/(1-p));  ds: dS / scale  — both factors are applied once at the end
      float prs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) prs[j] = ex2(fmaf(lse[j], -MRB_LOG2E, sv[8 * c + j]));
      if (EDGE) {
        asm volatile("; edge tile" ::: "memory");   // (keeps this block behind a real branch)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int qq = q0 + 16 * c + 8 * hi + j;
          bool ok = qq < p.Sq;
          if (CAUSAL) ok = ok && (key <= qq);
          prs[j] = ok ? prs[j] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * c + j;
        const float pr = prs[j];
        float kfr = keep_scale, prd = pr;
        if (DROP) {
          kfr = __builtin_bit_cast(float, keepm[j] & __builtin_bit_cast(int, keep_scale));
          prd = __builtin_bit_cast(float, keepm[j] & __builtin_bit_cast(int, pr));
        }
        pd[j] = prd;
        ds[j] = pr * (kfr * dpacc[r] - del[j]);
      }
      const bf16x8 pc = pack8(pd), sc = pack8(ds);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + T_BYTES + (t_off[mt][c] ^ (sub << 6))), pc, dv[mt], 0, 0, 0);
        dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(base + (t_off[mt][c] ^ (sub << 6))), sc, dk[mt], 0, 0, 0);
      }
    }
  };

  const int nst = (p.Sq + 63) >> 6;
  if (st0 < nst) stage(st0, 0);
  for (int st = st0, it = 0; st < nst; ++st, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nst) stage(st + 1, (it + 1) & 1);
    const char* base = sm + (it & 1) * STAGE;
    if (active) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {   // (sub is a compile-time constant here: two instances)
        const int q0 = st * 64 + 32 * sub;
        if (q0 < p.Sq && !(CAUSAL && q0 + 31 < kb0)) tile(q0 + 32 > p.Sq || (CAUSAL && q0 < kb0 + 31), q0, base, sub);
      }
    }
  }
  const float fk = key_ok ? p.scale : 0.f, fv = key_ok ? keep_scale : 0.f;  // masked keys: exact zeros
  if (key < p.Sk) {
    bf16_t* kp = const_cast<bf16_t*>(p.dK.ptr) + b * p.dK.bs + h * p.dK.hs + (long long)key * p.dK.rs;
    bf16_t* vp = const_cast<bf16_t*>(p.dV.ptr) + b * p.dV.bs + h * p.dV.hs + (long long)key * p.dV.rs;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = mt * 32 + 8 * g + 4 * hi;
        if (d0 < p.D) {
          *reinterpret_cast<uint2*>(kp + d0) = make_uint2(pack2bf(dk[mt][4 * g] * fk, dk[mt][4 * g + 1] * fk), pack2bf(dk[mt][4 * g + 2] * fk, dk[mt][4 * g + 3] * fk));
          *reinterpret_cast<uint2*>(vp + d0) = make_uint2(pack2bf(dv[mt][4 * g] * fv, dv[mt][4 * g + 1] * fv), pack2bf(dv[mt][4 * g + 2] * fv, dv[mt][4 * g + 3] * fv));
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
  a.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);   // 16-bit draws (attn_draw, v3)
  if (a.drop.seed_ptr && a.drop.thresh24 < 1u) a.drop.thresh24 = 1u;   // (the packed compare subtracts thresh - 1)
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

// the LDS kernels of the head_dim-64, non-causal, Sq > 32 shapes (T5 encoder, Q-Former... ): all three passes take the same
// decision, so the stored keep bits are either written and read or ignored by all of them
static bool use_lds64(const AttnArgs& a, int flags) {
  return a.D == 64 && !(flags & F_CAUSAL) && a.Sq > 32 && !((flags & F_MASK) && a.Skpad > 32 * ATTN_MASK_WORDS);   // (the LDS key-mask bitmap holds 8192 keys)
}

static int attn_ks2_mode() {  // MRB_ATTN_KS2: 0 = never, 1 = always, unset = when there are too few query tiles to give every SIMD four waves
  const char* e = getenv("MRB_ATTN_KS2");   // (read per launch: tests pin the choice while they compare runs of different batch sizes)
  return e ? atoi(e) : -1;
}
template <int FL>
static void fwd_lds64(const AttnArgs& a, dim3 grid, hipStream_t stream) {
  constexpr int LDS = 2 * (64 * 128 + 64 * 128) + 1040 + ((FL & F_MASK) ? 4 * ATTN_MASK_WORDS : 0);
  constexpr int LDS2 = 4 * (2 * 16 + 2) * 64 * 4 > LDS ? 4 * (2 * 16 + 2) * 64 * 4 : LDS;   // KS2: the merge area (4 x 34 x 64 floats) reuses the stages
  const long long qtiles = (long long)((a.Sq + 31) / 32) * a.H * a.B;
  const int mode = attn_ks2_mode();
  // (The same split for the dQ kernel was built and measured: it needs <= 128 VGPRs for two 8-wave blocks per CU, has 142, and forced
  // to 128 it spills 19 dwords: 246 vs 213 us per layer — not kept.  The dK/dV kernel holds 193.)
  // 1024 SIMDs: fewer than 3 query tiles (= waves) each -> split the keys (measured: 84.5 vs 97.4 us per layer; the key-mask variants
  // follow since their mask comes from the LDS bitmap: 148 -> 123 VGPRs, two 8-wave blocks per CU)
  const bool ks2 = mode == 1 || (mode < 0 && qtiles <= 3 * 1024 && a.Sk >= 256);
  if (ks2) {
    if ((FL & F_DROP) && a.dbits) hipLaunchKernelGGL((attn_fwd_lds_kernel<64, ((FL & F_DROP) ? (FL | F_DBITS) : FL) | F_KS2>), grid, dim3(512), LDS2, stream, a);
    else hipLaunchKernelGGL((attn_fwd_lds_kernel<64, FL | F_KS2>), grid, dim3(512), LDS2, stream, a);
    return;
  }
  if ((FL & F_DROP) && a.dbits) hipLaunchKernelGGL((attn_fwd_lds_kernel<64, (FL & F_DROP) ? (FL | F_DBITS) : FL>), grid, dim3(256), LDS, stream, a);
  else hipLaunchKernelGGL((attn_fwd_lds_kernel<64, FL>), grid, dim3(256), LDS, stream, a);
}
template <int FL>
static void dq_lds64(const AttnArgs& a, dim3 grid, hipStream_t stream) {
  constexpr int LDS = 2 * 3 * 64 * 128 + 1040 + ((FL & F_MASK) ? 4 * ATTN_MASK_WORDS : 0);
  if ((FL & F_DROP) && a.dbits) hipLaunchKernelGGL((attn_bwd_dq_lds_kernel<(FL & F_DROP) ? (FL | F_DBITS) : FL>), grid, dim3(256), LDS, stream, a);
  else hipLaunchKernelGGL((attn_bwd_dq_lds_kernel<FL>), grid, dim3(256), LDS, stream, a);
}
template <int FL>
static void dkv_lds64(const AttnArgs& a, dim3 grid, hipStream_t stream) {
  constexpr int LDS = 2 * (4 * 64 * 128 + 512 + 1024) + 1040;
  if ((FL & F_DROP) && a.dbits && !(FL & F_CAUSAL))
    hipLaunchKernelGGL((attn_bwd_dkv_lds_kernel<((FL & F_DROP) && !(FL & F_CAUSAL)) ? (FL | F_DBITS) : FL>), grid, dim3(256), LDS, stream, a);
  else hipLaunchKernelGGL((attn_bwd_dkv_lds_kernel<FL>), grid, dim3(256), LDS, stream, a);
}
static void fwd_lds96(const AttnArgs& a, dim3 grid, hipStream_t stream, bool f16 = false) {
  if (f16) hipLaunchKernelGGL((attn_fwd_lds_kernel<96, F_VROW | F_F16>), grid, dim3(256), 2 * (64 * 192 + 96 * 128) + 1040, stream, a);
  else if (a.V.ptr) hipLaunchKernelGGL((attn_fwd_lds_kernel<96, F_VROW>), grid, dim3(256), 2 * (64 * 192 + 96 * 128) + 1040, stream, a);
  else hipLaunchKernelGGL((attn_fwd_lds_kernel<96, 0>), grid, dim3(256), 2 * (64 * 192 + 96 * 128) + 1040, stream, a);
}

// ---- F_XS workspace (host-side, per calling thread; mrblip_attention_set_split_workspace): [tickets: 4096 uint32 (zero)][partials]
static thread_local void* g_xs_ws = nullptr;
static thread_local long long g_xs_bytes = 0;
static thread_local int g_xs_n = 0;
#define ATTN_XS_TICKETS 4096
extern "C" int mrblip_attention_set_split_workspace(void* ws, long long bytes, int n_split) {
  MRB_REQUIRE(!ws || (bytes >= ATTN_XS_TICKETS * 4 && ((uintptr_t)ws % 16) == 0 && n_split >= 0 && n_split <= 64), "attention split workspace: need >= 16 KB of 16-B aligned, ZEROED device memory and n_split <= 64");
  g_xs_ws = ws; g_xs_bytes = ws ? bytes : 0; g_xs_n = ws ? n_split : 0;
  return MRBLIP_OK;
}
// blocks per (batch, head) of the cross-block key split, 0 = not applicable: only the few-query form with a long key range and so few
// (batch, head) pairs that the chip is mostly idle (the T5 decoder's cross attention: 32 heads x 2012 keys; NOT the Q-Former's 720 pairs)
static int attn_xs_split(const AttnArgs& a, bool split) {
  if (!split || !g_xs_ws || a.D != 64 || a.B * a.H > 256 || a.Sk < 1024) return 0;
  const int tiles = (a.Sk + 31) / 32;
  // chunks of ~8 key tiles (two per wave), whatever the batch: a clip's result must not depend on how many clips share the launch
  // (B = 4 per GPU == four accumulated single-clip steps, tests/test_fullsize_gpu.py)
  int n = g_xs_n > 0 ? g_xs_n : (tiles + 7) / 8;
  if (n > tiles / 4) n = tiles / 4;                       // at least 4 key tiles (one per wave) per block
  if (n > 64) n = 64;
  const long long need = ATTN_XS_TICKETS * 4 + (long long)a.B * a.H * n * (9 * 64 * 16);
  if (n < 2 || need > g_xs_bytes) return 0;
  return n;
}

// the key-split form on 8 waves instead of 4 (MRB_ATTN_SPLIT8=1; measured in round 3 on the decoder's cross attention, 8-14 queries x 2012
// keys, one block per head: decoder phases 8.79 + 13.96 ms with 4 waves, 8.85 + 14.21 ms with 8, step 70.44 vs 70.52 ms — the block is
// bound by what ONE CU streams, not by the length of a wave's tile chain; off by default)
static bool attn_split8(const AttnArgs& a) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MRB_ATTN_SPLIT8"); on = (e && e[0] == '1') ? 1 : 0; }
  return on && a.Sk >= 1024;
}

template <int DP>
static int launch_fwd(const AttnArgs& a, int flags, hipStream_t stream) {
  const bool split = a.Sq <= 32 && a.Sk >= 256 && !(flags & F_CAUSAL);
  dim3 grid(split ? 1 : (a.Sq + 127) / 128, a.H, a.B);
  const int xs = DP == 64 ? attn_xs_split(a, split) : 0;
  AttnArgs ax = a;
  if (xs) { ax.xs_cnt = (uint32_t*)g_xs_ws; ax.xs_ws = (float*)((char*)g_xs_ws + ATTN_XS_TICKETS * 4); ax.xs_n = xs; }
#define X(DPV, FL)                                                                                             \
  if (flags == (FL)) {                                                                                         \
    if (xs && DPV == 64 && !((FL) & (F_CAUSAL | F_LUT))) hipLaunchKernelGGL((attn_fwd_kernel<64, ((FL) & ~(F_CAUSAL | F_LUT)) | F_SPLIT | F_XS>), dim3(xs, a.H, a.B), dim3(256), 0, stream, ax); \
    else if (split && !((FL) & F_CAUSAL) && DPV == 64 && attn_split8(a)) hipLaunchKernelGGL((attn_fwd_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT, DPV == 64 ? 8 : 4>), grid, dim3(DPV == 64 ? 512 : 256), 0, stream, a); \
    else if (split && !((FL) & F_CAUSAL)) hipLaunchKernelGGL((attn_fwd_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT>), grid, dim3(256), 0, stream, a); \
    else if (DPV == 64 && use_lds64(a, (FL))) fwd_lds64<((FL) & ~F_CAUSAL)>(a, grid, stream);                       \
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
  const int xs = DP == 64 ? attn_xs_split(a, split) : 0;
  AttnArgs ax = a;
  if (xs) { ax.xs_cnt = (uint32_t*)g_xs_ws; ax.xs_ws = (float*)((char*)g_xs_ws + ATTN_XS_TICKETS * 4); ax.xs_n = xs; }
#define X(DPV, FL)                                                                                             \
  if (flags == (FL)) {                                                                                         \
    if (xs && DPV == 64 && !((FL) & (F_CAUSAL | F_LUT))) hipLaunchKernelGGL((attn_bwd_dq_kernel<64, ((FL) & ~(F_CAUSAL | F_LUT)) | F_SPLIT | F_XS>), dim3(xs, a.H, a.B), dim3(256), 0, stream, ax); \
    else if (split && !((FL) & F_CAUSAL) && DPV == 64 && attn_split8(a)) hipLaunchKernelGGL((attn_bwd_dq_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT, DPV == 64 ? 8 : 4>), gq, dim3(DPV == 64 ? 512 : 256), 0, stream, a); \
    else if (split && !((FL) & F_CAUSAL)) hipLaunchKernelGGL((attn_bwd_dq_kernel<DPV, ((FL) & ~F_CAUSAL) | F_SPLIT>), gq, dim3(256), 0, stream, a); \
    else if (DPV == 64 && use_lds64(a, (FL))) dq_lds64<((FL) & ~F_CAUSAL)>(a, gq, stream);                          \
    else hipLaunchKernelGGL((attn_bwd_dq_kernel<DPV, (FL)>), gq, dim3(256), 0, stream, a);                      \
    if (DPV == 64 && a.D == 64 && a.Sq > 32) dkv_lds64<(FL)>(a, gk, stream);                                     \
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
                                    const uint32_t* seed_ptr, uint32_t site, float p_drop, uint32_t* drop_bits, hipStream_t stream) {
  AttnArgs a = {};
  a.dbits = drop_bits;
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
  if (Sq > 32) fwd_lds96(a, dim3((Sq + 127) / 128, H, B), stream);
  else hipLaunchKernelGGL((attn_fwd_kernel<96, 0>), dim3((Sq + 127) / 128, H, B), dim3(256), 0, stream, a);
  return mrblip_check_launch("attention_fwd");
}

// The plain (ViT) forward reading V ROW-major, i.e. straight from the fused qkv projection output: no mrblip_head_transpose of V.
// head_dim in (64, 96], Sq > 32, no bias / mask / dropout (eva_vit.py:128-145).  v_strides like q_strides.
extern "C" int mrblip_attention_fwd_rowv(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* V,
                                         const long long* v_strides, void* O, const long long* o_strides, float* LSE, int B, int H, int Sq,
                                         int Sk, int D, float scale, hipStream_t stream) {
  AttnArgs a = {};
  if (int e = attn_fill(a, Q, q_strides, K, k_strides, V, v_strides, B, H, Sq, Sk, D)) return e;
  MRB_REQUIRE(D > 64 && D <= 96 && Sq > 32, "attention_fwd_rowv: the row-major-V form covers head_dim in (64, 96] with more than 32 queries (the ViT)");
  MRB_REQUIRE(V != nullptr && (v_strides[2] % 8) == 0 && ((uintptr_t)V % 16) == 0, "attention_fwd_rowv: V rows must be 16-B aligned");
  a.O = T4{(const bf16_t*)O, o_strides[0], o_strides[1], o_strides[2]};
  a.LSE = LSE; a.scale = scale;
  fwd_lds96(a, dim3((Sq + 127) / 128, H, B), stream);
  return mrblip_check_launch("attention_fwd_rowv");
}

// the same on IEEE fp16 Q / K / V / O (the fp16-operand ViT: eva_vit.py:128-145 under fp16 autocast — fp16 matmuls, fp32 softmax)
extern "C" int mrblip_attention_fwd_rowv_f16(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* V,
                                             const long long* v_strides, void* O, const long long* o_strides, float* LSE, int B, int H, int Sq,
                                             int Sk, int D, float scale, hipStream_t stream) {
  AttnArgs a = {};
  if (int e = attn_fill(a, Q, q_strides, K, k_strides, V, v_strides, B, H, Sq, Sk, D)) return e;
  MRB_REQUIRE(D > 64 && D <= 96 && Sq > 32, "attention_fwd_rowv_f16: the row-major-V form covers head_dim in (64, 96] with more than 32 queries (the ViT)");
  MRB_REQUIRE(V != nullptr && (v_strides[2] % 8) == 0 && ((uintptr_t)V % 16) == 0, "attention_fwd_rowv_f16: V rows must be 16-B aligned");
  a.O = T4{(const bf16_t*)O, o_strides[0], o_strides[1], o_strides[2]};
  a.LSE = LSE; a.scale = scale;
  fwd_lds96(a, dim3((Sq + 127) / 128, H, B), stream, true);
  return mrblip_check_launch("attention_fwd_rowv_f16");
}

// Backward.  Needs the forward's O and LSE, the transposed copies Kt [B,H,DP,Skpad], Qt / dOt [B,H,DP,Sqpad];
// writes Delta [B,H,Sqpad] (scratch), dQ, dK, dV (bf16, strided like their forward tensors).
extern "C" int mrblip_attention_bwd(const void* Q, const long long* q_strides, const void* K, const long long* k_strides,
                                    const void* V, const long long* v_strides, const void* O, const long long* o_strides,
                                    const void* dO, const long long* do_strides, const void* Kt, const void* Qt, const void* dOt,
                                    const float* LSE, float* Delta, void* dQ, const long long* dq_strides, void* dK,
                                    const long long* dk_strides, void* dV, const long long* dv_strides, int B, int H, int Sq, int Sk,
                                    int D, float scale, const float* bias_lut, const int* kmask, int causal,
                                    const uint32_t* seed_ptr, uint32_t site, float p_drop, const uint32_t* drop_bits, hipStream_t stream) {
  AttnArgs a = {};
  if (int e = attn_fill(a, Q, q_strides, K, k_strides, V, v_strides, B, H, Sq, Sk, D)) return e;
  a.dbits = const_cast<uint32_t*>(drop_bits);
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
  tmp.drop.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);   // element dropout of the SOURCE (mrb_keep: 16-bit draws), not attention draws
  hipLaunchKernelGGL(head_transpose_kernel, dim3(Spad / 32, H, B), dim3(256), 0, stream, s, (bf16_t*)dst, S, D, DP, Spad, tmp.drop);
  return mrblip_check_launch("head_transpose");
}
