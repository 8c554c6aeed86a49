// Fused projection of the T5 DECODER rows (R <= 16 rows: 8-14 label tokens of one clip; modeling_t5.py:747-826 with peft LoRA Linear):
//
//     xin  = bf16(RMSNorm(x) * gamma)            (NORM inputs: q/k/v, EncDecAttention.q, wi_0/wi_1)   or the given bf16 rows (o, wo)
//     u    = bf16( dropout_lora(xin) (sA)^T )                                          [R, Rk]    (saved for the backward)
//     acc  = xin W^T + u B^T                                                           [R, N]
//     out  = bf16(acc) | residual + dropout(acc) (fp32) | bf16(dropout(gelu(acc_0) * acc_1)) with out2 = bf16([acc_0 | acc_1])
//
// and the backward's input gradient of the same layers (the same arithmetic with other operands):
//
//     g    = bf16( dy (sB) )   (= "u", no dropout),   dx = dy W + mask_lora (.) (g (sA))  (+ residual)       (ext_masked)
//
// Until round 3 each of these was two launches — the LoRA row kernel (or the fused RMSNorm + row kernel) and the skinny GEMM — on a
// chain of ~45 launches per decoder layer whose length, not whose work, sets the decoder's share of the train step (tools/prof_layer.py:
// 283 + 325 us of kernels per layer for 113 MB of weights; skipping 3.2 ms of that chain shortens the step by 1.9 ms).  One launch now:
// a block owns 16 * NT output columns, its 8 waves split K; every wave streams its share of the block's W rows, of the stacked LoRA "down"
// rows (32-96 KB, L2 resident, re-read by every block) and of the input rows as 16-B loads straight into v_mfma_f32_16x16x32_bf16
// fragments (lane (i = l & 15, kg = l >> 4) holds 8 consecutive k of row i), two batches of k-steps in flight; the partial accumulators
// meet in LDS in fixed order; the rank-Rk "up" product is one more MFMA on u rounded to bf16 exactly as the two-launch path stores it.
// NORM: the normalised rows are built once per block in LDS (padded rows: conflict-free fragment reads) and block 0 saves them.
// Dropout masks: the element indices and hashes of the kernels this replaces (lora.hip: pair hash over row * K + k; gemm.hip epilogue:
// single hash over row * N + n; the LoRA-backward mask: pair hash over row * N + n).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct DecProjArgs {
  const float* x32; long long ldx32; const float* gamma; float eps;   // NORM input (x32 != nullptr)
  bf16_t* xin; long long ldxin;                                       // bf16 [R, >= K]: the input rows, or (NORM) where the normalised rows are saved
  const bf16_t* W; long long ldw;                                     // [N (gated: 2 N), K]
  const bf16_t* A; long long lda; int Rk;                             // [Rk, K] stacked scaled LoRA "down" rows (backward: s B^T block-diagonal)
  const bf16_t* Bt; long long ldbt;                                   // [N (gated: 2 N), 64]: LoRA "up" rows (backward: (s A)^T)
  bf16_t* U; long long ldu;                                           // [R, 64] out: u (forward) / g (backward)
  int R, N, K;
  void* out; long long ldo; const float* residual; long long ldr; bf16_t* out2; long long ldo2;
  DropoutArg in_drop, out_drop, ext_drop;
  // mode 0 only: head-transposed copies of up to three consecutive column ranges of width t_inner (q | k | v of a fused projection, or one
  // range = the whole output): tout[j][b][h][d][s] = out[b * t_rows + s][j * t_inner + h * 64 + d] — what mrblip_head_transpose would
  // write, INCLUDING the zero pad columns [t_rows, t_spad) of every (clip, head, d) row, which the lane of the clip's last position clears on every call)
  bf16_t* tout[3]; int t_inner, t_rows, t_spad; long long t_bs, t_hs;
};

typedef uint32_t dp_u32x4 __attribute__((ext_vector_type(4)));

// RT: 16-row tiles of the input (R <= 16 RT).  RT > 1 — the S = 72 encoder of the 20-frame workload, the decoder rows of a 4-clip batch —
// keeps a k-step's weight fragments in registers and multiplies them with every row tile; input rows come from global memory only
// (no fused RMSNorm: 16 RT rows of 2048 bf16 do not fit the LDS), the partial sums go through the LDS buffer one row tile at a time.
template <int NT, int MODE, int UB, int RT = 1>   // NT tiles of 16 output columns (gated: NT of wi_0 and the same NT of wi_1); MODE 0 bf16, 1 fp32 (+ residual), 2 gated
__global__ __launch_bounds__(512) void dec_proj_kernel(const DecProjArgs p) {
  constexpr int NTW = MODE == 2 ? 2 * NT : NT, NA = 2, NACC = NTW + NA;
  static_assert(RT == 1 || MODE != 2, "row tiles: plain and residual outputs only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [red: 8 waves x NACC x 64 lanes x 16 B][ubuf: 16 rows x 64 B][NORM: 16 rows x (2 K + 16) B]
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  bf16_t* ubuf = reinterpret_cast<bf16_t*>(smem + 8 * NACC * 64 * 16);
  char* xs = smem + 8 * NACC * 64 * 16 + 1024;
  const bool NORM = p.x32 != nullptr;   // block-uniform
  const int RS = p.K * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * 16 * NT;
  const bool in_drop = p.in_drop.seed_ptr != nullptr, out_drop = p.out_drop.seed_ptr != nullptr, ext_masked = p.ext_drop.seed_ptr != nullptr;
  const uint32_t* sp = in_drop ? p.in_drop.seed_ptr : out_drop ? p.out_drop.seed_ptr : p.ext_drop.seed_ptr;
  const uint32_t seed = sp ? mrb_seed_load(sp) : 0u;

  // ---- main loop: this wave's k-steps of 32
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, (int)((((long long)(MODE == 2 ? 2 : 1) * p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((((long long)p.Rk - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xin, 0, NORM ? 0 : (int)((((long long)p.R - 1) * p.ldxin + p.K) * 2), 0x00020000);
  uint32_t woff[NTW], aoff[NA];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int col = n0 + (t % NT) * 16 + l15;                        // output column of this lane's W row
    const long long row = (MODE == 2 && t >= NT) ? (long long)p.N + col : col;
    woff[t] = col < p.N ? (uint32_t)((row * p.ldw + kg * 8) * 2) : 0x80000000u;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) aoff[a] = (uint32_t)(((long long)(16 * a + l15) * p.lda + kg * 8) * 2);   // rows >= Rk: beyond the resource
  uint32_t xoff[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) xoff[rt] = (uint32_t)(((long long)(16 * rt + l15) * p.ldxin + kg * 8) * 2);    // rows >= R: beyond the resource
  const int nks = p.K >> 5;
  const int per = (nks + 7) >> 3;
  const int ks0 = w * per, ks1 = min(nks, ks0 + per);
  f32x4 acc[RT][NTW], accu[RT][NA];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < NA; ++a) accu[rt][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  dp_u32x4 wf[2][UB][NTW], af[2][UB][NA], xf[2][UB][RT];
  auto fetch = [&](int buf, int ks) {   // unconditional (past the share: out-of-range offsets -> zeros, no traffic), see lora_thin_kernel
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const bool ok = ks + u < ks1;
      const uint32_t kb = (uint32_t)(ks + u) * 64u;
#pragma unroll
      for (int t = 0; t < NTW; ++t) wf[buf][u][t] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? woff[t] + kb : 0x80000000u, 0, 0);
#pragma unroll
      for (int a = 0; a < NA; ++a) af[buf][u][a] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? aoff[a] + kb : 0x80000000u, 0, 0);
      if (!NORM) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) xf[buf][u][rt] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xoff[rt] + kb : 0x80000000u, 0, 0);
      }
    }
  };
  auto consume = [&](int buf, int ks) {
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      dp_u32x4 x;
      if (NORM) {
        const int k = min(ks + u, nks - 1) * 32 + kg * 8;
        x = *reinterpret_cast<const dp_u32x4*>(xs + l15 * RS + k * 2);
        if (ks + u >= ks1) x = dp_u32x4{0u, 0u, 0u, 0u};
      } else {
        x = xf[buf][u][rt];
      }
      dp_u32x4 xd = x;
      if (in_drop) {   // block-uniform
        const uint32_t e = (uint32_t)(16 * rt + l15) * (uint32_t)p.K + (uint32_t)((ks + u) * 32 + kg * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bool k0, k1;
          mrb_keep2(e + 2 * q, seed, p.in_drop.site, p.in_drop.thresh24, k0, k1);
          xd[q] = (k0 ? x[q] & 0xffffu : 0u) | (k1 ? x[q] & 0xffff0000u : 0u);
        }
      }
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[buf][u][t]), __builtin_bit_cast(bf16x8, x), acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < NA; ++a)
        accu[rt][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[buf][u][a]), __builtin_bit_cast(bf16x8, xd), accu[rt][a], 0, 0, 0);
    }
  };
  fetch(0, ks0);
  if (NORM) {   // rows 2 w, 2 w + 1: RMSNorm -> bf16 rows in LDS (the arithmetic of norm_fwd_kernel<true>: v * rstd * gamma, one rounding);
                // both rows' loads are in flight together, beside the first batch of weight fragments requested above
    const int nv = p.K >> 2;
    float4 v[2][8];
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * w + i;
      const float4* xr = reinterpret_cast<const float4*>(p.x32 + (long long)(r < p.R ? r : 0) * p.ldx32);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j, cc = c < nv ? c : 0;
        v[i][j] = xr[cc];
      }
    }
    float4 gq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      gq[j] = reinterpret_cast<const float4*>(p.gamma)[c < nv ? c : 0];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * w + i;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        if (c >= nv || r >= p.R) v[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        q[i] += v[i][j].x * v[i][j].x + v[i][j].y * v[i][j].y + v[i][j].z * v[i][j].z + v[i][j].w * v[i][j].w;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 2 * w + i;
      const float rstd = rsqrtf(wave_sum(q[i]) / (float)p.K + p.eps);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        if (c < nv) {
          const float4 g = gq[j];
          const uint2 o = make_uint2(pack2bf(v[i][j].x * rstd * g.x, v[i][j].y * rstd * g.y), pack2bf(v[i][j].z * rstd * g.z, v[i][j].w * rstd * g.w));
          *reinterpret_cast<uint2*>(xs + r * RS + c * 8) = o;
          if (blockIdx.x == 0 && r < p.R) *reinterpret_cast<uint2*>(p.xin + (long long)r * p.ldxin + c * 4) = o;
        }
      }
    }
    __syncthreads();
  }

#pragma unroll 1
  for (int ks = ks0; ks < ks1; ks += 2 * UB) {
    fetch(1, ks + UB);
    consume(0, ks);
    fetch(0, ks + 2 * UB);
    if (ks + UB < ks1) consume(1, ks + UB);
  }

  // ---- the 8 partial sums meet in LDS (fixed order), one 16-row tile at a time
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
  const int rbase = 16 * rt;
  if (rt > 0) __syncthreads();   // the previous row tile's readers are done with red / ubuf
#pragma unroll
  for (int t = 0; t < NTW; ++t) red[(w * NACC + t) * 64 + lane] = acc[rt][t];
#pragma unroll
  for (int a = 0; a < NA; ++a) red[(w * NACC + NTW + a) * 64 + lane] = accu[rt][a];
  __syncthreads();
  if (w < NA) {   // u tile w: lane (r = l15, kg) holds j = 16 w + 4 kg .. + 3
    f32x4 v = red[(0 * NACC + NTW + w) * 64 + lane];
#pragma unroll
    for (int j = 1; j < 8; ++j) v += red[(j * NACC + NTW + w) * 64 + lane];
    const float post = in_drop ? p.in_drop.inv_keep : 1.0f;
    const uint2 ub = make_uint2(pack2bf(v[0] * post, v[1] * post), pack2bf(v[2] * post, v[3] * post));
    const int j0 = 16 * w + 4 * kg;
    *reinterpret_cast<uint2*>(ubuf + l15 * 32 + j0) = ub;
    if (blockIdx.x == 0 && rbase + l15 < p.R && j0 < p.Rk) *reinterpret_cast<uint2*>(p.U + (long long)(rbase + l15) * p.ldu + j0) = ub;
  }
  __syncthreads();
  if (w < NT) {   // output tile w (gated: the pair w, w + NT)
    const int col = n0 + w * 16;
    constexpr int NH = MODE == 2 ? 2 : 1;
    f32x4 h[NH];
    const dp_u32x4 uf = *reinterpret_cast<const dp_u32x4*>(ubuf + l15 * 32 + kg * 8);    // u[r = l15][8 kg .. + 7]
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Bt), 0, (int)((((long long)(MODE == 2 ? 2 : 1) * p.N - 1) * p.ldbt + 64) * 2), 0x00020000);
#pragma unroll
    for (int s = 0; s < NH; ++s) {
      const int t = w + s * NT;
      f32x4 v = red[(0 * NACC + t) * 64 + lane];
#pragma unroll
      for (int j = 1; j < 8; ++j) v += red[(j * NACC + t) * 64 + lane];
      const long long brow = (long long)(s ? p.N : 0) + col + l15;
      const dp_u32x4 bf = __builtin_amdgcn_raw_buffer_load_b128(rb, (col + l15 < p.N) ? (uint32_t)((brow * p.ldbt + kg * 8) * 2) : 0x80000000u, 0, 0);
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      if (ext_masked) {   // backward: dx = dy W + mask (.) (g A): the rank-Rk product on its own, masked per element (pair hash over row * N + n)
        f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, uf), zero, 0, 0, 0);
        const uint32_t idx = (uint32_t)(rbase + l15) * (uint32_t)p.N + (uint32_t)(col + 4 * kg);
        bool k0, k1, k2, k3;
        mrb_keep2(idx, seed, p.ext_drop.site, p.ext_drop.thresh24, k0, k1);
        mrb_keep2(idx + 2, seed, p.ext_drop.site, p.ext_drop.thresh24, k2, k3);
        v[0] += k0 ? e[0] * p.ext_drop.inv_keep : 0.f; v[1] += k1 ? e[1] * p.ext_drop.inv_keep : 0.f;
        v[2] += k2 ? e[2] * p.ext_drop.inv_keep : 0.f; v[3] += k3 ? e[3] * p.ext_drop.inv_keep : 0.f;
      } else {
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf), __builtin_bit_cast(bf16x8, uf), v, 0, 0, 0);
      }
      h[s] = v;
    }
    const int r = rbase + l15, n = col + 4 * kg;   // this lane: row r, columns n .. n + 3
    if (r < p.R && n < p.N) {
      if (MODE == 2) {
        if (p.out2) {   // (generation keeps no pre-activations)
          *reinterpret_cast<uint2*>(p.out2 + (long long)r * p.ldo2 + n) = make_uint2(pack2bf(h[0][0], h[0][1]), pack2bf(h[0][2], h[0][3]));
          *reinterpret_cast<uint2*>(p.out2 + (long long)r * p.ldo2 + p.N + n) = make_uint2(pack2bf(h[NH - 1][0], h[NH - 1][1]), pack2bf(h[NH - 1][2], h[NH - 1][3]));
        }
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          float g0 = h[0][i], g1 = h[0][i + 1];
          gelu_erf2(g0, g1);
          y[i] = g0 * h[NH - 1][i];
          y[i + 1] = g1 * h[NH - 1][i + 1];
        }
        if (out_drop) {
          bool kq0, kq1, kq2, kq3;   // (n % 4 == 0, N even: two pair hashes)
          mrb_keep4((uint32_t)r * (uint32_t)p.N + (uint32_t)n, seed, p.out_drop.site, p.out_drop.thresh24, kq0, kq1, kq2, kq3);
          y[0] = kq0 ? y[0] * p.out_drop.inv_keep : 0.f;
          y[1] = kq1 ? y[1] * p.out_drop.inv_keep : 0.f;
          y[2] = kq2 ? y[2] * p.out_drop.inv_keep : 0.f;
          y[3] = kq3 ? y[3] * p.out_drop.inv_keep : 0.f;
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)r * p.ldo + n) = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
      } else {
        float y[4] = {h[0][0], h[0][1], h[0][2], h[0][3]};
        if (out_drop) {
          bool kq0, kq1, kq2, kq3;   // (n % 4 == 0, N even: two pair hashes)
          mrb_keep4((uint32_t)r * (uint32_t)p.N + (uint32_t)n, seed, p.out_drop.site, p.out_drop.thresh24, kq0, kq1, kq2, kq3);
          y[0] = kq0 ? y[0] * p.out_drop.inv_keep : 0.f;
          y[1] = kq1 ? y[1] * p.out_drop.inv_keep : 0.f;
          y[2] = kq2 ? y[2] * p.out_drop.inv_keep : 0.f;
          y[3] = kq3 ? y[3] * p.out_drop.inv_keep : 0.f;
        }
        if (MODE == 1) {
          if (p.residual) {
            const float4 q = *reinterpret_cast<const float4*>(p.residual + (long long)r * p.ldr + n);
            y[0] += q.x; y[1] += q.y; y[2] += q.z; y[3] += q.w;
          }
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)r * p.ldo + n) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
          const uint2 ob = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)r * p.ldo + n) = ob;
          if (p.t_inner > 0) {   // block-uniform
            const int which = n / p.t_inner;
            bf16_t* td = which < 3 ? p.tout[which] : nullptr;
            if (td) {
              const int c = n - which * p.t_inner, hh = c >> 6, d0 = c & 63, bb = r / p.t_rows, ss = r - bb * p.t_rows;
              bf16_t* q = td + bb * p.t_bs + hh * p.t_hs + (long long)d0 * p.t_spad + ss;
              q[0] = (bf16_t)(ob.x & 0xffffu); q[p.t_spad] = (bf16_t)(ob.x >> 16);
              q[2 * p.t_spad] = (bf16_t)(ob.y & 0xffffu); q[3 * p.t_spad] = (bf16_t)(ob.y >> 16);
              // the pad columns [t_rows, t_spad) of the (clip, head, d) rows this lane owns: zeroed on EVERY call by the lane of the
              // clip's last position, like head_transpose does — the workspaces are capacity-based views (engine.buf), so when the label
              // length or the batch changes between steps a tile may hold another layout's values there
              if (ss == p.t_rows - 1) {
                for (int z = 1; z < p.t_spad - ss; ++z) { q[z] = 0; q[p.t_spad + z] = 0; q[2 * p.t_spad + z] = 0; q[3 * p.t_spad + z] = 0; }
              }
            }
          }
        }
      }
    }
  }
  }
}

// ---- round 4: the same projection as a STREAMING kernel (R <= 16 rows).  What the round-3 kernel above costs is not its arithmetic but its
// shape: one 32-column tile per block, so a block is launch + one memory round trip + reduction + a chain of dependent epilogue loads
// (8-9 GB/s per CU; qkv: 192 blocks = THREE rounds on the 64 CUs the look-ahead ViT leaves to the decoder: 15 us alone, 49 us in the
// step).  Here a block owns a CONTIGUOUS RANGE of 16-column tiles (grid = min(tiles, mrblip_dec_proj_set_grid) blocks) and streams their
// weight rows back to back: the work that does not depend on the tile — the RMSNorm of the rows (or the copy of the bf16 rows) into
// LDS and the LoRA "down" product u — is done ONCE per block, the next tile's weight fragments are requested before this tile's partial
// sums meet (two batches of UB k-steps per wave always in flight, across tile boundaries), the rank-Rk "up" fragment and the residual of
// a tile are requested a whole tile ahead, and the reduction buffer alternates so that a tile costs ONE barrier.  K split over the 8
// waves and the summation order are those of the kernel above: bit-identical results (tests/test_kernels_gpu.py::test_dec_proj_*).
//   XL: K <= 2048 — the input rows live in LDS (NORM: normalised there; else copied once); longer K streams the rows with the weights.
static thread_local int g_dec_grid = 0;   // blocks of the streaming kernel (0: the CU count)
static int g_dec_v2 = -1;                 // 1: streaming kernel for R <= 16 (default), 0: the one-tile-per-block kernel of round 3 (MRB_DEC_PROJ_V2=0)
// n_blocks > 0: the streaming kernel's grid for the calling thread's later launches (0: one block per CU; < 0: unchanged) — the engine asks
// for as many blocks as the look-ahead ViT leaves CUs, so that a projection is ONE round of resident blocks.  version 0 / 1 selects the
// kernel (any other value: unchanged).  Returns the previous n_blocks.
extern "C" int mrblip_dec_proj_config(int n_blocks, int version) {
  const int prev = g_dec_grid;
  if (n_blocks >= 0) g_dec_grid = n_blocks;
  if (version == 0 || version == 1) g_dec_v2 = version;
  return prev;
}

#ifndef DP2_UB_LONGK
#define DP2_UB_LONGK 4   // k-steps per batch of the K > 2048 forms (input rows streamed with the weights)
#endif
template <int MODE, int XM, int UB>   // XM: 0 = input rows streamed with the weights (K > 2048), 1 = bf16 rows copied into LDS, 2 = RMSNorm of fp32 rows into LDS
__global__ __launch_bounds__(512) void dec_proj2_kernel(const DecProjArgs p, const int tpb) {
  constexpr int NTW = MODE == 2 ? 2 : 1, NA = 2;
  constexpr bool XL = XM != 0, NORM = XM == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [red: 2 x 8 waves x NTW x 64 lanes x 16 B][ured: 8 x NA x 64 x 16 B][ubuf: 16 rows x 64 B][XL: 16 rows x (2 K + 16) B]
  f32x4* red = reinterpret_cast<f32x4*>(smem);
  f32x4* ured = reinterpret_cast<f32x4*>(smem + 2 * 8 * NTW * 1024);
  bf16_t* ubuf = reinterpret_cast<bf16_t*>(smem + 2 * 8 * NTW * 1024 + 8 * NA * 1024);
  char* xs = smem + 2 * 8 * NTW * 1024 + 8 * NA * 1024 + 1024;
  const int RS = p.K * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kg = lane >> 4;
  const int ntiles = (p.N + 15) >> 4;
  const int tile0 = blockIdx.x * tpb, tile1 = min(ntiles, tile0 + tpb);
  const bool in_drop = p.in_drop.seed_ptr != nullptr, out_drop = p.out_drop.seed_ptr != nullptr, ext_masked = p.ext_drop.seed_ptr != nullptr;
  const uint32_t* sp = in_drop ? p.in_drop.seed_ptr : out_drop ? p.out_drop.seed_ptr : p.ext_drop.seed_ptr;
  const uint32_t seed = sp ? mrb_seed_load(sp) : 0u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, (int)((((long long)(MODE == 2 ? 2 : 1) * p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((((long long)p.Rk - 1) * p.lda + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xin, 0, NORM ? 0 : (int)((((long long)p.R - 1) * p.ldxin + p.K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Bt), 0, (int)((((long long)(MODE == 2 ? 2 : 1) * p.N - 1) * p.ldbt + 64) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.residual), 0, (MODE == 1 && p.residual) ? (int)((((long long)p.R - 1) * p.ldr + p.N) * 4) : 0, 0x00020000);
  const int nks = p.K >> 5;
  const int per = (nks + 7) >> 3;
  const int ks0 = w * per, ks1 = min(nks, ks0 + per);
  const int nb = (per + UB - 1) / UB;                 // batches per tile (the same for every wave)
  const uint32_t lane_koff = (uint32_t)(kg * 16);
  const uint32_t xoff = (uint32_t)(((long long)l15 * p.ldxin) * 2) + lane_koff;    // rows >= R: beyond the resource

  // ---- the weight stream: item (tile, batch); fetches run two items ahead of the multiplications
  dp_u32x4 wf[2][UB][NTW], xf[2][UB];
  int ft = tile0, fb = 0;                              // next item to fetch
  auto fetch = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const int col = ft * 16 + l15;
    const bool tv = ft < tile1 && col < p.N;
    uint32_t wrow[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) wrow[t] = (uint32_t)(((long long)(t ? p.N + col : col) * p.ldw) * 2) + lane_koff;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int ks = ks0 + fb * UB + u;
      const bool ok = tv && ks < ks1;
      const uint32_t kb = (uint32_t)ks * 64u;
#pragma unroll
      for (int t = 0; t < NTW; ++t) wf[buf][u][t] = __builtin_amdgcn_raw_buffer_load_b128(rw, ok ? wrow[t] + kb : 0x80000000u, 0, 0);
      if (!XL) xf[buf][u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xoff + kb : 0x80000000u, 0, 0);
    }
    if (++fb == nb) { fb = 0; ++ft; }
  };
  // ---- loads first, in the order their data is needed: the input rows (NORM: fp32 rows + gamma; else the bf16 rows), the first two batches of
  // the thin operand, then two batches of weight fragments and the first tile's epilogue operands.  (vmcnt retires in issue order: a wait
  // for the rows must not stand behind the weight stream.)
  const int nv = p.K >> 2;     // NORM: float4 per row
  const int nc = p.K >> 3;     // copy: 16-B chunks per row (<= 256)
  float4 v[2][8], gq[8];
  dp_u32x4 c4[2][4];
  if (XL) {
    if (NORM) {   // rows 2 w, 2 w + 1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 2 * w + i;
        const float4* xr = reinterpret_cast<const float4*>(p.x32 + (long long)(r < p.R ? r : 0) * p.ldx32);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = lane + 64 * j, cc = c < nv ? c : 0;
          v[i][j] = xr[cc];
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        gq[j] = reinterpret_cast<const float4*>(p.gamma)[c < nv ? c : 0];
      }
    } else {      // the bf16 rows as they are (rows >= R: beyond the resource -> zeros; chunks past the row: the last chunk again)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = min(lane + 64 * j, nc - 1);
          c4[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (uint32_t)(((long long)(2 * w + i) * p.ldxin) * 2) + (uint32_t)c * 16u, 0, 0);
        }
    }
  }
  constexpr int UA = 4;                               // k-steps per batch of the thin operand
  const int nba = (per + UA - 1) / UA;
  uint32_t aoff[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) aoff[a] = (uint32_t)(((long long)(16 * a + l15) * p.lda) * 2) + lane_koff;   // rows >= Rk: beyond the resource
  dp_u32x4 af[2][UA][NA], ax[2][UA];
  auto afetch = [&](auto bufc, int b) {
    constexpr int buf = decltype(bufc)::value;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int ks = ks0 + b * UA + u;
      const bool ok = ks < ks1;
      const uint32_t kb = (uint32_t)ks * 64u;
#pragma unroll
      for (int a = 0; a < NA; ++a) af[buf][u][a] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? aoff[a] + kb : 0x80000000u, 0, 0);
      if (!XL) ax[buf][u] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xoff + kb : 0x80000000u, 0, 0);
    }
  };
  afetch(std::integral_constant<int, 0>{}, 0);
  afetch(std::integral_constant<int, 1>{}, 1);
  fetch(std::integral_constant<int, 0>{});
  if (!XL) fetch(std::integral_constant<int, 1>{});   // (XL: the second batch goes out once the row registers are free)
  // wave 0: a tile's LoRA "up" fragment(s) and residual are requested when the tile BEFORE it starts (other waves, absent operands and
  // tiles past the block's range: out-of-range offsets -> zeros, no traffic)
  dp_u32x4 bfr[NTW], bfn[NTW];
  dp_u32x4 resq = {0u, 0u, 0u, 0u}, resn = {0u, 0u, 0u, 0u};
  auto epi_fetch = [&](int tile) {
    const int col = tile * 16 + l15;
    const bool v0 = w == 0 && tile < tile1 && col < p.N;
#pragma unroll
    for (int s = 0; s < NTW; ++s)
      bfn[s] = __builtin_amdgcn_raw_buffer_load_b128(rb, v0 ? (uint32_t)(((long long)(s ? p.N + col : col) * p.ldbt + kg * 8) * 2) : 0x80000000u, 0, 0);
    if (MODE == 1) {
      const int n = tile * 16 + 4 * kg;
      resn = __builtin_amdgcn_raw_buffer_load_b128(rr, (w == 0 && tile < tile1 && l15 < p.R && n < p.N) ? (uint32_t)(((long long)l15 * p.ldr + n) * 4) : 0x80000000u, 0, 0);
    }
  };
  epi_fetch(tile0);

  // ---- once per block: the input rows into LDS (XL), then u = bf16(dropout(x) (sA)^T)
  if (XL) {
    if (NORM) {   // the arithmetic of norm_fwd_kernel<true>: v * rstd * gamma, one rounding
      float q[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 2 * w + i;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = lane + 64 * j;
          if (c >= nv || r >= p.R) v[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
          q[i] += v[i][j].x * v[i][j].x + v[i][j].y * v[i][j].y + v[i][j].z * v[i][j].z + v[i][j].w * v[i][j].w;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = 2 * w + i;
        const float rstd = rsqrtf(wave_sum(q[i]) / (float)p.K + p.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = lane + 64 * j;
          if (c < nv) {
            const float4 g = gq[j];
            const uint2 o = make_uint2(pack2bf(v[i][j].x * rstd * g.x, v[i][j].y * rstd * g.y), pack2bf(v[i][j].z * rstd * g.z, v[i][j].w * rstd * g.w));
            *reinterpret_cast<uint2*>(xs + r * RS + c * 8) = o;
            if (blockIdx.x == 0 && r < p.R) *reinterpret_cast<uint2*>(p.xin + (long long)r * p.ldxin + c * 4) = o;
          }
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = min(lane + 64 * j, nc - 1);
          *reinterpret_cast<dp_u32x4*>(xs + (2 * w + i) * RS + c * 16) = c4[i][j];
        }
    }
    fetch(std::integral_constant<int, 1>{});
    __syncthreads();
  }
  auto x_lds = [&](int ks) -> dp_u32x4 {   // XL: this lane's 8 k of row l15 at k-step ks (zero past the wave's share: the weights are zero there,
                                           // but whatever lies behind the rows must not reach the MFMA as a NaN pattern)
    const int k = min(ks, nks - 1) * 32 + kg * 8;
    dp_u32x4 x = *reinterpret_cast<const dp_u32x4*>(xs + l15 * RS + k * 2);
    if (ks >= ks1) x = dp_u32x4{0u, 0u, 0u, 0u};
    return x;
  };
  {
    f32x4 accu[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) accu[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto aconsume = [&](auto bufc, int b) {
      constexpr int buf = decltype(bufc)::value;
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        const int ks = ks0 + b * UA + u;
        dp_u32x4 x = XL ? x_lds(ks) : ax[buf][u];
        if (in_drop) {   // block-uniform
          const uint32_t e = (uint32_t)l15 * (uint32_t)p.K + (uint32_t)(ks * 32 + kg * 8);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bool k0, k1;
            mrb_keep2(e + 2 * q, seed, p.in_drop.site, p.in_drop.thresh24, k0, k1);
            x[q] = (k0 ? x[q] & 0xffffu : 0u) | (k1 ? x[q] & 0xffff0000u : 0u);
          }
        }
#pragma unroll
        for (int a = 0; a < NA; ++a)
          accu[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[buf][u][a]), __builtin_bit_cast(bf16x8, x), accu[a], 0, 0, 0);
      }
    };
#pragma unroll 1
    for (int b = 0; b < nba; b += 2) {
      aconsume(std::integral_constant<int, 0>{}, b);
      afetch(std::integral_constant<int, 0>{}, b + 2);
      if (b + 1 < nba) aconsume(std::integral_constant<int, 1>{}, b + 1);
      afetch(std::integral_constant<int, 1>{}, b + 3);
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) ured[(w * NA + a) * 64 + lane] = accu[a];
    __syncthreads();
    if (w < NA) {   // u tile w: lane (r = l15, kg) holds j = 16 w + 4 kg .. + 3
      f32x4 v = ured[(0 * NA + w) * 64 + lane];
#pragma unroll
      for (int j = 1; j < 8; ++j) v += ured[(j * NA + w) * 64 + lane];
      const float post = in_drop ? p.in_drop.inv_keep : 1.0f;
      const uint2 ub = make_uint2(pack2bf(v[0] * post, v[1] * post), pack2bf(v[2] * post, v[3] * post));
      const int j0 = 16 * w + 4 * kg;
      *reinterpret_cast<uint2*>(ubuf + l15 * 32 + j0) = ub;
      if (blockIdx.x == 0 && l15 < p.R && j0 < p.Rk) *reinterpret_cast<uint2*>(p.U + (long long)l15 * p.ldu + j0) = ub;
    }
    __syncthreads();
  }
  const dp_u32x4 uf = *reinterpret_cast<const dp_u32x4*>(ubuf + l15 * 32 + kg * 8);    // u[r = l15][8 kg .. + 7]

  // ---- the tiles
  f32x4 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ct = tile0, cb = 0, par = 0;                     // item being multiplied; reduction buffer of its tile
  auto tile_done = [&]() {
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      red[((par * 8 + w) * NTW + t) * 64 + lane] = acc[t];
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (w == 0) {
      const int col = ct * 16;
      f32x4 h[NTW];
#pragma unroll
      for (int s = 0; s < NTW; ++s) {
        f32x4 v = red[((par * 8 + 0) * NTW + s) * 64 + lane];
#pragma unroll
        for (int j = 1; j < 8; ++j) v += red[((par * 8 + j) * NTW + s) * 64 + lane];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (ext_masked) {   // backward: dx = dy W + mask (.) (g A): the rank-Rk product on its own, masked per element (pair hash over row * N + n)
          f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[s]), __builtin_bit_cast(bf16x8, uf), zero, 0, 0, 0);
          const uint32_t idx = (uint32_t)l15 * (uint32_t)p.N + (uint32_t)(col + 4 * kg);
          bool k0, k1, k2, k3;
          mrb_keep2(idx, seed, p.ext_drop.site, p.ext_drop.thresh24, k0, k1);
          mrb_keep2(idx + 2, seed, p.ext_drop.site, p.ext_drop.thresh24, k2, k3);
          v[0] += k0 ? e[0] * p.ext_drop.inv_keep : 0.f; v[1] += k1 ? e[1] * p.ext_drop.inv_keep : 0.f;
          v[2] += k2 ? e[2] * p.ext_drop.inv_keep : 0.f; v[3] += k3 ? e[3] * p.ext_drop.inv_keep : 0.f;
        } else {
          v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[s]), __builtin_bit_cast(bf16x8, uf), v, 0, 0, 0);
        }
        h[s] = v;
      }
      const int r = l15, n = col + 4 * kg;   // this lane: row r, columns n .. n + 3
      if (r < p.R && n < p.N) {
        if (MODE == 2) {
          if (p.out2) {   // (generation keeps no pre-activations)
            *reinterpret_cast<uint2*>(p.out2 + (long long)r * p.ldo2 + n) = make_uint2(pack2bf(h[0][0], h[0][1]), pack2bf(h[0][2], h[0][3]));
            *reinterpret_cast<uint2*>(p.out2 + (long long)r * p.ldo2 + p.N + n) = make_uint2(pack2bf(h[NTW - 1][0], h[NTW - 1][1]), pack2bf(h[NTW - 1][2], h[NTW - 1][3]));
          }
          float y[4];
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            float g0 = h[0][i], g1 = h[0][i + 1];
            gelu_erf2(g0, g1);
            y[i] = g0 * h[NTW - 1][i];
            y[i + 1] = g1 * h[NTW - 1][i + 1];
          }
          if (out_drop) {
            bool kq0, kq1, kq2, kq3;   // (n % 4 == 0, N even: two pair hashes)
            mrb_keep4((uint32_t)r * (uint32_t)p.N + (uint32_t)n, seed, p.out_drop.site, p.out_drop.thresh24, kq0, kq1, kq2, kq3);
            y[0] = kq0 ? y[0] * p.out_drop.inv_keep : 0.f;
            y[1] = kq1 ? y[1] * p.out_drop.inv_keep : 0.f;
            y[2] = kq2 ? y[2] * p.out_drop.inv_keep : 0.f;
            y[3] = kq3 ? y[3] * p.out_drop.inv_keep : 0.f;
          }
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)r * p.ldo + n) = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
        } else {
          float y[4] = {h[0][0], h[0][1], h[0][2], h[0][3]};
          if (out_drop) {
            bool kq0, kq1, kq2, kq3;   // (n % 4 == 0, N even: two pair hashes)
            mrb_keep4((uint32_t)r * (uint32_t)p.N + (uint32_t)n, seed, p.out_drop.site, p.out_drop.thresh24, kq0, kq1, kq2, kq3);
            y[0] = kq0 ? y[0] * p.out_drop.inv_keep : 0.f;
            y[1] = kq1 ? y[1] * p.out_drop.inv_keep : 0.f;
            y[2] = kq2 ? y[2] * p.out_drop.inv_keep : 0.f;
            y[3] = kq3 ? y[3] * p.out_drop.inv_keep : 0.f;
          }
          if (MODE == 1) {
            const f32x4 q = __builtin_bit_cast(f32x4, resq);   // zeros without a residual
            y[0] += q[0]; y[1] += q[1]; y[2] += q[2]; y[3] += q[3];
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long long)r * p.ldo + n) = make_float4(y[0], y[1], y[2], y[3]);
          } else {
            const uint2 ob = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (long long)r * p.ldo + n) = ob;
            if (p.t_inner > 0) {   // block-uniform: the head-transposed copies (see the kernel above)
              const int which = n / p.t_inner;
              bf16_t* td = which < 3 ? p.tout[which] : nullptr;
              if (td) {
                const int c = n - which * p.t_inner, hh = c >> 6, d0 = c & 63, bb = r / p.t_rows, ss = r - bb * p.t_rows;
                bf16_t* q = td + bb * p.t_bs + hh * p.t_hs + (long long)d0 * p.t_spad + ss;
                q[0] = (bf16_t)(ob.x & 0xffffu); q[p.t_spad] = (bf16_t)(ob.x >> 16);
                q[2 * p.t_spad] = (bf16_t)(ob.y & 0xffffu); q[3 * p.t_spad] = (bf16_t)(ob.y >> 16);
                if (ss == p.t_rows - 1) {
                  for (int z = 1; z < p.t_spad - ss; ++z) { q[z] = 0; q[p.t_spad + z] = 0; q[2 * p.t_spad + z] = 0; q[3 * p.t_spad + z] = 0; }
                }
              }
            }
          }
        }
      }
    }
    par ^= 1;
    ++ct;
  };
  auto consume = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    if (cb == 0) {
#pragma unroll
      for (int s = 0; s < NTW; ++s) bfr[s] = bfn[s];
      resq = resn;
      epi_fetch(ct + 1);
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int ks = ks0 + cb * UB + u;
      const dp_u32x4 x = XL ? x_lds(ks) : xf[buf][u];
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[buf][u][t]), __builtin_bit_cast(bf16x8, x), acc[t], 0, 0, 0);
    }
    if (++cb == nb) { cb = 0; tile_done(); }
  };
  const int items = (tile1 - tile0) * nb;
#pragma unroll 1
  for (int it = 0; it < items; it += 2) {
    consume(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 0>{});
    if (it + 1 < items) consume(std::integral_constant<int, 1>{});
    fetch(std::integral_constant<int, 1>{});
  }
}

static void dp_drop(DropoutArg& d, const uint32_t* seed_ptr, uint32_t site, float p) {
  d.seed_ptr = (p > 0.f) ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p);
}

// mode: 0 = bf16 out, 1 = fp32 out (+ residual), 2 = gated (W / Bt hold 2 N rows: wi_0 then wi_1; out2 [R, 2 N], optional, receives the pre-activations).
// x32 != NULL: RMSNorm(x32) * gamma is the input (K <= 2048) and is saved to xin; else xin is the bf16 input.  ext_p > 0: the LoRA-backward
// form (the rank-Rk product masked per element before it is added).  R <= 16 (<= 80 without the fused RMSNorm / gating), N % 16 == 0, K % 32 == 0, Rk <= 32 and a multiple of 8.
extern "C" int mrblip_dec_proj(const float* x32, long long ldx32, const float* gamma, float eps, void* xin, long long ldxin, const void* W,
                               long long ldw, const void* A, long long lda, int Rk, const void* Bt, long long ldbt, void* U, long long ldu, int R,
                               int N, int K, int mode, void* out, long long ldo, const float* residual, long long ldr, void* out2, long long ldo2,
                               const uint32_t* seed_ptr, uint32_t in_site, float in_p, uint32_t out_site, float out_p, uint32_t ext_site, float ext_p,
                               void* tout0, void* tout1, void* tout2, int t_inner, int t_rows, int t_spad, long long t_bs, long long t_hs,
                               hipStream_t stream) {
  MRB_REQUIRE(R > 0 && R <= 80 && N > 0 && (N % 16) == 0 && K > 0 && (K % 32) == 0, "dec_proj: need R <= 80, N %% 16 == 0, K %% 32 == 0 (R=%d N=%d K=%d)", R, N, K);
  MRB_REQUIRE(R <= 16 || (!x32 && mode != 2), "dec_proj: more than 16 rows only with bf16 input rows and a plain / residual output (R=%d)", R);
  MRB_REQUIRE(Rk > 0 && Rk <= 32 && (Rk % 8) == 0 && ldu >= Rk && (ldu % 4) == 0, "dec_proj: bad LoRA rank rows (Rk=%d)", Rk);
  MRB_REQUIRE(mode >= 0 && mode <= 2 && W && A && Bt && U && xin && out, "dec_proj: bad mode / missing operand");
  MRB_REQUIRE(!x32 || (K <= 2048 && gamma && (ldx32 % 4) == 0), "dec_proj: the fused RMSNorm takes K <= 2048");
  MRB_REQUIRE((ldw % 8) == 0 && (lda % 8) == 0 && (ldbt % 8) == 0 && (ldxin % 8) == 0 && (ldo % 4) == 0 && ((uintptr_t)W % 16) == 0 &&
                  ((uintptr_t)A % 16) == 0 && ((uintptr_t)Bt % 16) == 0 && ((uintptr_t)xin % 16) == 0 && ((uintptr_t)out % 16) == 0,
              "dec_proj: 16-B alignment");
  MRB_REQUIRE((long long)(mode == 2 ? 2 : 1) * N * ldw * 2 < (1ll << 31), "dec_proj: W exceeds the 2 GiB buffer range");
  MRB_REQUIRE(!((in_p > 0.f || out_p > 0.f || ext_p > 0.f) && !seed_ptr), "dec_proj: dropout needs a device seed pointer");
  DecProjArgs a;
  a.x32 = x32; a.ldx32 = ldx32; a.gamma = gamma; a.eps = eps; a.xin = (bf16_t*)xin; a.ldxin = ldxin; a.W = (const bf16_t*)W; a.ldw = ldw;
  a.A = (const bf16_t*)A; a.lda = lda; a.Rk = Rk; a.Bt = (const bf16_t*)Bt; a.ldbt = ldbt; a.U = (bf16_t*)U; a.ldu = ldu; a.R = R; a.N = N; a.K = K;
  a.out = out; a.ldo = ldo; a.residual = residual; a.ldr = ldr; a.out2 = (bf16_t*)out2; a.ldo2 = ldo2;
  const bool any_t = tout0 || tout1 || tout2;
  MRB_REQUIRE(!any_t || (mode == 0 && t_inner > 0 && (t_inner % 64) == 0 && t_rows > 0 && t_spad >= t_rows), "dec_proj: head-transposed copies need mode 0 and a head layout");
  a.tout[0] = (bf16_t*)tout0; a.tout[1] = (bf16_t*)tout1; a.tout[2] = (bf16_t*)tout2;
  a.t_inner = any_t ? t_inner : 0; a.t_rows = t_rows; a.t_spad = t_spad; a.t_bs = t_bs; a.t_hs = t_hs;
  dp_drop(a.in_drop, seed_ptr, in_site, in_p);
  dp_drop(a.out_drop, seed_ptr, out_site, out_p);
  dp_drop(a.ext_drop, seed_ptr, ext_site, ext_p);
  // round 4: R <= 16 rows take the streaming kernel (MRB_DEC_PROJ_V2=0: the one-tile-per-block kernel of round 3)
  if (g_dec_v2 < 0) { const char* e = getenv("MRB_DEC_PROJ_V2"); g_dec_v2 = (e && e[0] == '0') ? 0 : 1; }
  if (g_dec_v2 && R <= 16) {
    static int ncu = 0, env_grid = -1;
    if (ncu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (env_grid < 0) { const char* e = getenv("MRB_DEC_GRID"); env_grid = e ? atoi(e) : 0; }
    const int G = env_grid > 0 ? env_grid : (g_dec_grid > 0 ? g_dec_grid : ncu);
    const int tiles = (N + 15) / 16;
    const int tpb = (tiles + G - 1) / G;
    const int grid2 = (tiles + tpb - 1) / tpb;
    const bool xl = K <= 2048;
    const int ntw2 = mode == 2 ? 2 : 1;
    const int LDS2 = 2 * 8 * ntw2 * 1024 + 8 * 2 * 1024 + 1024 + (xl ? 16 * (K * 2 + 16) : 0);
    static bool attr2[9] = {};
#define MRB_DP2_LAUNCH(ID, MODE_, XM_, UB_)                                                                                       \
  {                                                                                                                                \
    auto k = dec_proj2_kernel<MODE_, XM_, UB_>;                                                                                    \
    if (!attr2[ID]) {                                                                                                              \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 2 * 1024 + 8 * 2 * 1024 + 1024 + 16 * (2048 * 2 + 16)) != hipSuccess) { \
        mrblip_set_error("dec_proj: cannot raise dynamic LDS");                                                                    \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr2[ID] = true;                                                                                                            \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid2), dim3(512), LDS2, stream, a, tpb);                                                           \
  }
    const int xm = !xl ? 0 : (x32 ? 2 : 1);
    if (mode == 2) { if (xm == 2) MRB_DP2_LAUNCH(0, 2, 2, 4) else if (xm == 1) MRB_DP2_LAUNCH(1, 2, 1, 4) else MRB_DP2_LAUNCH(2, 2, 0, 4) }
    else if (mode == 0) { if (xm == 2) MRB_DP2_LAUNCH(3, 0, 2, 8) else if (xm == 1) MRB_DP2_LAUNCH(4, 0, 1, 8) else MRB_DP2_LAUNCH(5, 0, 0, DP2_UB_LONGK) }
    else { if (xm == 2) MRB_DP2_LAUNCH(6, 1, 2, 8) else if (xm == 1) MRB_DP2_LAUNCH(7, 1, 1, 8) else MRB_DP2_LAUNCH(8, 1, 0, DP2_UB_LONGK) }
#undef MRB_DP2_LAUNCH
    return mrblip_check_launch("dec_proj");
  }
  // 32 output columns per block; 16 (twice the blocks, half the weight bytes each) for the long-K projections of a 2048-wide output, whose
  // 64 blocks are otherwise a serial stream of 20-40 k-steps per wave (MRB_DEC_PROJ_NT1=0 keeps 32)
  static int nt1 = -1;
  if (nt1 < 0) { const char* e = getenv("MRB_DEC_PROJ_NT1"); nt1 = (e && e[0] == '0') ? 0 : 1; }
  const int rt = R <= 16 ? 1 : R <= 48 ? 3 : 5;
  const bool narrow = nt1 && mode != 2 && K >= 4096 && N <= 4096 && rt == 1;
  const int nt = narrow ? 1 : 2;
  const int grid = (N + 16 * nt - 1) / (16 * nt);
  const int ntw = mode == 2 ? 2 * nt : nt;
  const int LDS = 8 * (ntw + 2) * 64 * 16 + 1024 + (x32 ? 16 * (K * 2 + 16) : 0);
  static bool attr[16] = {};
#define MRB_DP_LAUNCH(ID, NT_, MODE_, UB_, RT_)                                                                                    \
  {                                                                                                                                \
    auto k = dec_proj_kernel<NT_, MODE_, UB_, RT_>;                                                                                \
    if (!attr[ID]) {                                                                                                               \
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * (2 * NT_ + 2) * 64 * 16 + 1024 + 16 * (2048 * 2 + 16)) != hipSuccess) { \
        mrblip_set_error("dec_proj: cannot raise dynamic LDS");                                                                    \
        return MRBLIP_ELAUNCH;                                                                                                     \
      }                                                                                                                            \
      attr[ID] = true;                                                                                                             \
    }                                                                                                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDS, stream, a);                                                                  \
  }
  if (mode == 2) MRB_DP_LAUNCH(0, 2, 2, 2, 1)
  else if (mode == 0) {
    if (rt == 1) { if (narrow) MRB_DP_LAUNCH(1, 1, 0, 4, 1) else MRB_DP_LAUNCH(2, 2, 0, 4, 1) }
    else if (rt == 3) MRB_DP_LAUNCH(3, 2, 0, 2, 3)
    else MRB_DP_LAUNCH(4, 2, 0, 1, 5)
  } else {
    if (rt == 1) { if (narrow) MRB_DP_LAUNCH(5, 1, 1, 4, 1) else MRB_DP_LAUNCH(6, 2, 1, 4, 1) }
    else if (rt == 3) MRB_DP_LAUNCH(7, 2, 1, 2, 3)
    else MRB_DP_LAUNCH(8, 2, 1, 1, 5)
  }
#undef MRB_DP_LAUNCH
  return mrblip_check_launch("dec_proj");
}
