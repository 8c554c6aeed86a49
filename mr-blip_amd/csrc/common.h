// Shared device/host helpers for libmrblip_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MRBLIP_OK 0
#define MRBLIP_EINVAL (-1)
#define MRBLIP_ELAUNCH (-2)

void mrblip_set_error(const char* fmt, ...);
int mrblip_check_launch(const char* what);

// Workspace of the ORDERED reductions (cross-entropy row terms, bias column sums, LayerNorm dgamma / dbeta): partial sums leave the blocks
// with write-through stores and the block that draws the last ticket adds them in a fixed order, so the result is the same bits on every
// run.  The scratch and the tickets are CALLER-provided (mrblip_set_reduce_workspace, per calling thread; round 6 — until then they were
// library-owned __device__ arrays, i.e. process-global state that two streams would have corrupted): zeroed device memory of
// MRB_RWS_BYTES, one per stream that may run these kernels concurrently.  Without one the kernels fall back to fp32 atomics (arrival order).
#define MRB_RWS_CE_ROWS 4096
#define MRB_RWS_CS_MAXY 64
#define MRB_RWS_CS_MAXN 8192
#define MRB_RWS_DW_MAXB 256
#define MRB_RWS_DW_MAXD 2048
#define MRB_RWS_TICKET_CE 0                                        /* uint32 index */
#define MRB_RWS_TICKET_DW 1
#define MRB_RWS_TICKET_CS 4                                        /* .. 4 + MRB_RWS_CS_MAXN / 256 */
#define MRB_RWS_OFF_CE 1024ll                                      /* byte offsets */
#define MRB_RWS_OFF_CS (MRB_RWS_OFF_CE + 4ll * MRB_RWS_CE_ROWS)
#define MRB_RWS_OFF_DW (MRB_RWS_OFF_CS + 4ll * MRB_RWS_CS_MAXY * MRB_RWS_CS_MAXN)
#define MRB_RWS_BYTES (MRB_RWS_OFF_DW + 4ll * MRB_RWS_DW_MAXB * 2 * MRB_RWS_DW_MAXD)
char* mrblip_reduce_workspace();   // the calling thread's registered workspace (>= MRB_RWS_BYTES) or nullptr

#define MRB_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      mrblip_set_error(__VA_ARGS__);      \
      return MRBLIP_EINVAL;               \
    }                                     \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, like torch .bfloat16())
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float -> bf16 through the native type: hipcc lowers the conversion to v_cvt_pk_bf16_f32 (RNE, 2 elements per instruction)
// instead of ~7 integer VALU ops per element of a hand-written rounding.
typedef __bf16 mrb_bf16v2 __attribute__((ext_vector_type(2)));
typedef float mrb_f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const mrb_f32v2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, mrb_bf16v2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }
// ---- IEEE fp16 operands (round 4): the reference's GPU arithmetic for the frozen ViT is fp16 autocast (blip2_mr.py:446, fp16 weights
// eva_vit.py:397-412, 439-441) — 3 more mantissa bits than bf16 at the same MFMA rate.  Same 16-bit containers, other bit meaning.
typedef _Float16 mrb_f16v2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {   // round-to-nearest-even, like torch .half()
  const mrb_f32v2 f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, mrb_f16v2));
}
template <bool F16>
__device__ __forceinline__ uint32_t pack2x(float lo, float hi) { return F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
// 32x32x16 MFMA on 16-bit fragments held as 8 shorts: bf16 or fp16 by template flag (same rate)
template <bool F16>
__device__ __forceinline__ f32x16 mfma32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- counter-based dropout RNG.  One 32-bit hash per element; the oracle restates it in numpy
// (oracle/mrblip_oracle.py: dropout_keep) so training-mode parity can be checked with p > 0.
// seed = *seed_ptr (device memory, bumped once per step so hipGraph replays draw fresh masks), site = call-site id.
__device__ __forceinline__ uint32_t mrb_hash(uint32_t idx, uint32_t seed, uint32_t site) {
  // two multiply-xorshift rounds (integer multiplies are quarter rate on CDNA: keep them to two)
  uint32_t h = (idx ^ seed) * 0x9E3779B1u + site * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0xC2B2AE3Du;
  h ^= h >> 13;
  return h;
}
// One 32-bit hash serves the element PAIR (idx & ~1, idx | 1): even index -> low 16 bits, odd index -> high 16 bits.
// keep iff the 16-bit draw >= thresh16 = round(p * 65536)   (p = 0.1 -> 6554/65536 = 0.10001)
// s_waitcnt vmcnt(0) as an INSTRUCTION the compiler's wait-count pass sees (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15).  Placed between a
// kernel's last load and a run of conditional stores it tells the pass that nothing is pending: without it every conditional block gets its
// own conservative vmcnt(0), which also waits for the previous block's STORES to be acknowledged.
#define MRB_ALL_LOADS_DONE() __builtin_amdgcn_s_waitcnt(0x0F70)

// the step's dropout seed lives in device memory (written by the host-side engine before the step, never inside a kernel that reads it):
// fetch it with ONE scalar load (constant address space -> s_load_dword through the scalar cache, which is invalidated at every kernel
// start) instead of a per-use vector load
__device__ __forceinline__ uint32_t mrb_seed_load(const uint32_t* p) {
  return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ bool mrb_keep(uint32_t idx, uint32_t seed, uint32_t site, uint32_t thresh16) {
  const uint32_t h = mrb_hash(idx >> 1, seed, site);
  return ((idx & 1u) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}
// both draws of the pair starting at the EVEN index idx
__device__ __forceinline__ void mrb_keep2(uint32_t idx_even, uint32_t seed, uint32_t site, uint32_t thresh16, bool& k0, bool& k1) {
  const uint32_t h = mrb_hash(idx_even >> 1, seed, site);
  k0 = (h & 0xffffu) >= thresh16;
  k1 = (h >> 16) >= thresh16;
}

// the draws of FOUR consecutive elements starting at an EVEN index e: two pair hashes.  (Four mrb_keep(e + i) calls cost four: the compiler
// cannot know that e is even, so it cannot merge the hashes of e and e + 1.)  Every caller's e is row * (even row length) + (column % 4 == 0).
__device__ __forceinline__ void mrb_keep4(uint32_t e, uint32_t seed, uint32_t site, uint32_t thresh16, bool& k0, bool& k1, bool& k2, bool& k3) {
  mrb_keep2(e, seed, site, thresh16, k0, k1);
  mrb_keep2(e + 2u, seed, site, thresh16, k2, k3);
}

// LINEAR form of the counter hash for kernels that walk consecutive indices (attention-probability dropout): hash(idx) =
// fin(idx * MRB_H1 + base), base = mrb_lin_base(seed, site).  idx * MRB_H1 is a Weyl sequence in idx, so a kernel evaluates it with
// adds (a lane constant + a wave-uniform term computed on the scalar unit + a compile-time constant) and pays ONE quarter-rate
// integer multiply per hash — the finaliser's — instead of mrb_hash's two.  Restated in the oracle (dropout_hash_lin).
#define MRB_H1 0x9E3779B1u
__device__ __forceinline__ uint32_t mrb_lin_base(uint32_t seed, uint32_t site) { return seed * MRB_H1 + site * 0x85EBCA77u; }
__device__ __forceinline__ uint32_t mrb_lin_fin(uint32_t t) {
  t ^= t >> 15;
  t *= 0xC2B2AE3Du;
  t ^= t >> 13;
  return t;
}

// Round 5: the finaliser of the attention-probability draws on the FULL-RATE 24-bit multiplier (v_mul_u32_u24: low 32 bits of the product
// of the operands' low 24 bits; v_mul_lo_u32 is quarter rate).  The xor-shift in front folds bits 15..31 into the 24 bits the multiplier
// sees; statistics of the resulting draws (marginals, neighbour conditionals, quad patterns): tests/test_host_cpu.py.
// The output is a function of 24 bits, so a (seed, site) has at most 2^24 distinct quad hashes: one T5-XL encoder layer at S = 2012 draws
// 32.4 M quads, 87 % of which share their hash with some other quad of the layer (14.6 M distinct values, no value more than 10 times, never
// twice inside one query row: test_attention_dropout_quad_hashes_repeat_at_production_size...).  Far-apart score tiles may therefore carry the
// same 4-key keep pattern; marginals, within-quad conditionals and row / neighbour independence are unaffected.  Folding the dropped top
// byte back in would cost 2 VALU per hash (~4 % of the VALU-bound encoder attention forward): not taken.
__device__ __forceinline__ uint32_t mrb_lin_fin24(uint32_t t) {
  t ^= t >> 15;
  t = (t & 0xffffffu) * 0x9E3779u;
  t ^= t >> 13;
  return t;
}

struct DropoutArg {
  const uint32_t* seed_ptr;  // nullptr or p == 0 -> disabled
  uint32_t site;
  uint32_t thresh24;  // draw threshold (historic field name): round(p * 65536) — 16-bit draws everywhere (element dropout: mrb_keep;
                      // attention probabilities: attention.hip "draws v3")
  float inv_keep;  // 1 / (1 - p)
};

// erf(x) ~ clamp(t * P(t^2), -1, 1), t = clamp(x, -3, 3), P of degree 7 in t^2 (iteratively re-weighted least-squares minimax fit,
// max abs error 8.1e-5 in fp32 => exact-erf GELU to 1.6e-4 abs, far below the bf16 resolution of the activations it feeds).
// Pure FMA / v_med3: no v_rcp / v_exp (quarter rate); the 2-wide form lowers to v_pk_fma_f32 / v_pk_mul_f32 (two elements per
// instruction) and each clamp is ONE v_med3_f32 — the GELU epilogue of the ViT fc1 GEMM is VALU time that no MFMA overlaps
// (tools/w4_stamps.py: 11 us of a 57 us tile with the previous degree-8 / min+max form).
typedef float mrb_f2 __attribute__((ext_vector_type(2)));
#define MRB_ERF_L 3.0f
#define MRB_ERF_HORNER(P, T2)                                                                           \
  P = -4.055320281e-07f;                                                                                \
  P = P * T2 + 1.715970865e-05f;    P = P * T2 + (-3.145938746e-04f); P = P * T2 + 3.318701084e-03f;   \
  P = P * T2 + (-2.268576619e-02f); P = P * T2 + 1.077177701e-01f;    P = P * T2 + (-3.732313514e-01f); \
  P = P * T2 + 1.127895752e+00f;
__device__ __forceinline__ float fast_erf(float x) {
  const float t = __builtin_amdgcn_fmed3f(x, -MRB_ERF_L, MRB_ERF_L);
  const float t2 = t * t;
  float p;
  MRB_ERF_HORNER(p, t2)
  return __builtin_amdgcn_fmed3f(t * p, -1.0f, 1.0f);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }
// two elements at once (packed fp32 math)
__device__ __forceinline__ void gelu_erf2(float& a, float& b) {
  mrb_f2 x = {a, b};
  mrb_f2 t = x * 0.70710678118654752440f;
  t[0] = __builtin_amdgcn_fmed3f(t[0], -MRB_ERF_L, MRB_ERF_L);
  t[1] = __builtin_amdgcn_fmed3f(t[1], -MRB_ERF_L, MRB_ERF_L);
  const mrb_f2 t2 = t * t;
  mrb_f2 p;
  MRB_ERF_HORNER(p, t2)
  mrb_f2 r = t * p;
  r[0] = __builtin_amdgcn_fmed3f(r[0], -1.0f, 1.0f);
  r[1] = __builtin_amdgcn_fmed3f(r[1], -1.0f, 1.0f);
  const mrb_f2 h = x * 0.5f;
  const mrb_f2 y = h * r + h;
  a = y[0];
  b = y[1];
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float kInvSqrt2Pi = 0.39894228040143267794f;
  return 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// wave64 sum with DPP adds only (no LDS crossbar): four in-row butterfly steps (quad_perm, quad_perm, row_half_mirror, row_mirror: every
// lane then holds its 16-lane row's sum), row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3 — lane 63 holds the total, read
// back as a wave-uniform scalar.  ~6 dependent VALU ops against 6 dependent ds_bpermute round trips of wave_sum().
#define MRB_DPP_STEP(V, CTRL, ROWMASK) \
  V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, ROWMASK, 0xf, false))
__device__ __forceinline__ float wave_sum_uniform(float v) {
  MRB_DPP_STEP(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
  MRB_DPP_STEP(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
  MRB_DPP_STEP(v, 0x141, 0xf);  // row_half_mirror
  MRB_DPP_STEP(v, 0x140, 0xf);  // row_mirror
  MRB_DPP_STEP(v, 0x142, 0xa);  // row_bcast:15 -> rows 1, 3
  MRB_DPP_STEP(v, 0x143, 0xc);  // row_bcast:31 -> rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
