// HBM-bound side kernels of the Mr. BLIP hot path: frame patchify (coalesced reads of the [BT,3,H,W] frame
// tensor), cls/pos assembly, indexed row copy (embedding gather + frame/timestamp interleave scatter and its
// transpose for the backward), 32->1 frame-token mean pool, activation backward, casts/dropout, cross entropy,
// AdamW.  All vectorised (8-16 B/lane) where the layout allows.
#include "common.h"
#define NEG_INF_F (-3.0e38f)

// ---------------------------------------------------------------------------------------------------------
// patchify: video fp32 [F,3,IMG,IMG] (or uint8, normalised on the fly) -> 16-bit [F*G*G, Kpad], row = (f, gy, gx),
// col = c*P*P + py*P + px (the Conv2d(k=s=P) weight order, eva_vit.py:196-203), cols >= 3*P*P zero.
// One block per (f, gy) patch row.  Its 3 x P image rows are read once with 16-B (fp32) / 4-B (uint8) loads, fully coalesced; its
// output — the G patch rows (f, gy, 0..G-1) — is ONE contiguous G x Kpad chunk, which is assembled in LDS in output order and then
// written with 16-B coalesced stores.  (Round 4: the first form stored every element with its own 2-byte global store, 0.30 of the
// HBM peak; P = 14 is not a multiple of the load width, so the scatter has to happen somewhere: in LDS it is free.)
// F16: the output is IEEE fp16 (fp16-operand ViT) instead of bf16.
template <bool U8, bool F16, int PC>   // PC: compile-time patch size (14 = ViT-g/14: the divisions by P become multiplies), 0 = run-time P
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ img_, bf16_t* __restrict__ out, int IMG, int P_, int G, int Kpad,
                                                       float m0, float m1, float m2, float s0, float s1, float s2) {
  extern __shared__ __attribute__((aligned(16))) char pf_sm[];
  bf16_t* tile = reinterpret_cast<bf16_t*>(pf_sm);            // [G][Kpad]
  const int P = PC ? PC : P_;
  const int f = blockIdx.y, gy = blockIdx.x;
  const int K = 3 * P * P, W4 = IMG >> 2;                      // 4-pixel groups per image row (IMG % 4 == 0 checked by the host)
  const int n_out16 = G * Kpad / 8;                            // 16-B pieces of the output chunk
  const int n_grp = 3 * P * W4;
  // all of this thread's loads first (independent, in flight together), then the pad zero fill, then the scatter into the tile
  constexpr int MAXG = 12;                                     // groups per thread the registers hold: 3 * 14 * 56 / 256 = 9.2 for ViT-g/14
  float4 q[MAXG];
  uint32_t wq[MAXG];
#pragma unroll
  for (int j = 0; j < MAXG; ++j) {
    const int g = threadIdx.x + 256 * j;
    if (g < n_grp) {
      const int rowi = g / W4, x0 = (g - rowi * W4) * 4;
      const int c = rowi / P, py = rowi - c * P;
      const long long off = (((long long)f * 3 + c) * IMG + gy * P + py) * IMG + x0;
      if (U8) wq[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(img_) + off);
      else q[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(img_) + off);
    }
  }
  for (int i = threadIdx.x; i < n_out16; i += 256) {           // zero the 16-B pieces that hold pad columns (K .. Kpad)
    const int col = (i * 8) % Kpad;
    if (col + 8 > K) reinterpret_cast<uint4*>(tile)[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  auto scatter = [&](int g, const float* v) {
    const int rowi = g / W4, x0 = (g - rowi * W4) * 4;
    const int c = rowi / P, py = rowi - c * P;
    const int cbase = c * P * P + py * P;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x0 + e, gx = x / P, px = x - gx * P;
      tile[gx * Kpad + cbase + px] = (bf16_t)(pack2x<F16>(v[e], 0.f) & 0xffffu);
    }
  };
#pragma unroll
  for (int j = 0; j < MAXG; ++j) {
    const int g = threadIdx.x + 256 * j;
    if (g < n_grp) {
      float v[4];
      if (U8) {
        const int c = (g / W4) / P;
        const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, stdv = c == 0 ? s0 : c == 1 ? s1 : s2;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)((wq[j] >> (8 * e)) & 0xffu), 255.0f), mean), stdv);
      } else {
        v[0] = q[j].x; v[1] = q[j].y; v[2] = q[j].z; v[3] = q[j].w;
      }
      scatter(g, v);
    }
  }
  for (int g = threadIdx.x + 256 * MAXG; g < n_grp; g += 256) {   // (strips larger than the register batch: generic tail)
    const int rowi = g / W4, x0 = (g - rowi * W4) * 4;
    const int c = rowi / P, py = rowi - c * P;
    const long long off = (((long long)f * 3 + c) * IMG + gy * P + py) * IMG + x0;
    float v[4];
    if (U8) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(img_) + off);
      const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, stdv = c == 0 ? s0 : c == 1 ? s1 : s2;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)((w >> (8 * e)) & 0xffu), 255.0f), mean), stdv);
    } else {
      const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(img_) + off);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    scatter(g, v);
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out + ((long long)(f * G + gy) * G) * Kpad);
  for (int i = threadIdx.x; i < n_out16; i += 256) dst[i] = reinterpret_cast<const uint4*>(tile)[i];
}

// x[f,0,:] = cls + pos[0]; x[f,1+p,:] = patch[f*NP+p,:] + pos[1+p]   (eva_vit.py:328-331), fp32 residual stream
__global__ __launch_bounds__(256) void vit_assemble_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, float* __restrict__ x, int NP, int D) {
  const int f = blockIdx.y, t = blockIdx.x;  // t in [0, NP]
  const float4* src = (t == 0) ? reinterpret_cast<const float4*>(cls) : reinterpret_cast<const float4*>(patch + ((long long)f * NP + t - 1) * D);
  const float4* ps = reinterpret_cast<const float4*>(pos + (long long)t * D);
  float4* dst = reinterpret_cast<float4*>(x + ((long long)f * (NP + 1) + t) * D);
  for (int i = threadIdx.x; i < D / 4; i += 256) {
    const float4 a = src[i], b = ps[i];
    dst[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}

// dst[dst_idx[i], :] (=|+=) src[src_idx[i], :]; src_idx < 0 writes zeros (left zero-padding, blip2_mr.py:744-753)
template <bool ACCUM>
__global__ __launch_bounds__(256) void row_copy_kernel(const float* __restrict__ src, long long lds_, const int* __restrict__ src_idx,
                                                       float* __restrict__ dst, long long ldd, const int* __restrict__ dst_idx, int D) {
  const int i = blockIdx.x;
  const int si = src_idx[i], di = dst_idx[i];
  if (di < 0) return;
  float4* d = reinterpret_cast<float4*>(dst + (long long)di * ldd);
  const float4* s = reinterpret_cast<const float4*>(src + (long long)(si < 0 ? 0 : si) * lds_);
  for (int j = threadIdx.x; j < D / 4; j += 256) {
    float4 v = (si < 0) ? make_float4(0.f, 0.f, 0.f, 0.f) : s[j];
    if (ACCUM) { const float4 o = d[j]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    d[j] = v;
  }
}

// mean over the n tokens of a frame (blip2_mr.py:493-498).  One wave per (frame, 128-column slab): lane = (token half,
// float4 column); 16 in-register adds per half, then ONE wavefront shuffle (lane ^ 32) combines the halves.
__global__ __launch_bounds__(64) void mean_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int D) {
  const int f = blockIdx.y, lane = threadIdx.x, th = lane >> 5, c4 = blockIdx.x * 32 + (lane & 31);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < D) {
    for (int t = th; t < n; t += 2) {
      const float4 v = reinterpret_cast<const float4*>(x + ((long long)f * n + t) * D)[c4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  acc.x += __shfl_xor(acc.x, 32, 64); acc.y += __shfl_xor(acc.y, 32, 64);
  acc.z += __shfl_xor(acc.z, 32, 64); acc.w += __shfl_xor(acc.w, 32, 64);
  if (th == 0 && c4 * 4 < D) {
    const float inv = 1.0f / (float)n;
    reinterpret_cast<float4*>(out + (long long)f * D)[c4] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}
__global__ __launch_bounds__(256) void mean_pool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int n, int D) {
  const int f = blockIdx.y, t = blockIdx.x;
  const float inv = 1.0f / (float)n;
  for (int i = threadIdx.x; i < D / 4; i += 256) {
    const float4 v = reinterpret_cast<const float4*>(dout + (long long)f * D)[i];
    reinterpret_cast<float4*>(dx + ((long long)f * n + t) * D)[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
  }
}

// fp32 -> (dropout) -> bf16 and/or fp32.   idx = row * ncols + col (same convention as the GEMM epilogue)
__global__ __launch_bounds__(256) void cast_drop_kernel(const float* __restrict__ x, long long ldx, bf16_t* out_b, long long ldob,
                                                        float* out_f, long long ldof, int M, int N, DropoutArg drop) {
  const long long total4 = (long long)M * (N / 4);
  const uint32_t seed = drop.seed_ptr ? *drop.seed_ptr : 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (N / 4)), c = (int)(i % (N / 4)) * 4;
    float4 v = *reinterpret_cast<const float4*>(x + (long long)m * ldx + c);
    if (drop.seed_ptr) {
      const uint32_t base = (uint32_t)m * (uint32_t)N + (uint32_t)c;
      bool k0, k1, k2, k3;   // (N % 4 == 0: base is even)
      mrb_keep4(base, seed, drop.site, drop.thresh24, k0, k1, k2, k3);
      v.x = k0 ? v.x * drop.inv_keep : 0.f;
      v.y = k1 ? v.y * drop.inv_keep : 0.f;
      v.z = k2 ? v.z * drop.inv_keep : 0.f;
      v.w = k3 ? v.w * drop.inv_keep : 0.f;
    }
    if (out_f) *reinterpret_cast<float4*>(out_f + (long long)m * ldof + c) = v;
    if (out_b) *reinterpret_cast<uint2*>(out_b + (long long)m * ldob + c) = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
  }
}

// dh = dy * gelu'(h)   (Qformer.py:349-360 backward), bf16 in/out, 8 elements per thread
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ h, bf16_t* __restrict__ dh, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const bf16x8 a = reinterpret_cast<const bf16x8*>(dy)[i], b = reinterpret_cast<const bf16x8*>(h)[i];
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f((bf16_t)a[j]) * gelu_erf_grad(bf2f((bf16_t)b[j]));
    uint4 r = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    reinterpret_cast<uint4*>(dh)[i] = r;
  }
}

// gated-GELU backward (modeling_t5.py:323-329): y = drop(gelu(h0) * h1);  h = [h0 | h1] ([M, 2*Nh]), dh same layout
// dy_ext (round 5, mrblip_gated_gelu_bwd_parts): a second bf16 part of dy that is added under the mask of ext_drop over [M, Nh] — the LoRA
// term g A of the wo projection's input gradient, written as its own part by mrblip_gemm_ksplit (mask = wo's lora_dropout keep mask of y)
__global__ __launch_bounds__(256) void gated_bwd_kernel(const bf16_t* __restrict__ dy, long long lddy, const bf16_t* __restrict__ h, long long ldh,
                                                        bf16_t* __restrict__ dh, long long lddh, int M, int Nh, DropoutArg drop,
                                                        const bf16_t* __restrict__ dy_ext = nullptr, DropoutArg ext_drop = DropoutArg{nullptr, 0u, 0u, 1.0f}) {
  const long long total8 = (long long)M * (Nh / 8);
  const uint32_t seed = drop.seed_ptr ? *drop.seed_ptr : ext_drop.seed_ptr ? *ext_drop.seed_ptr : 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (Nh / 8)), c = (int)(i % (Nh / 8)) * 8;
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + (long long)m * lddy + c);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(h + (long long)m * ldh + c);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(h + (long long)m * ldh + Nh + c);
    float d0[8], d1[8], ge[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool kp[8] = {true, true, true, true, true, true, true, true};
    if (drop.seed_ptr) {   // (c % 8 == 0, Nh % 8 == 0: element pairs share a hash)
#pragma unroll
      for (int j = 0; j < 8; j += 2) mrb_keep2((uint32_t)m * (uint32_t)Nh + (uint32_t)(c + j), seed, drop.site, drop.thresh24, kp[j], kp[j + 1]);
    }
    if (dy_ext) {
      const bf16x8 e = *reinterpret_cast<const bf16x8*>(dy_ext + (long long)m * lddy + c);
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        bool k0 = true, k1 = true;
        if (ext_drop.seed_ptr) mrb_keep2((uint32_t)m * (uint32_t)Nh + (uint32_t)(c + j), seed, ext_drop.site, ext_drop.thresh24, k0, k1);
        ge[j] = k0 ? bf2f((bf16_t)e[j]) * ext_drop.inv_keep : 0.f;
        ge[j + 1] = k1 ? bf2f((bf16_t)e[j + 1]) * ext_drop.inv_keep : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float gy = bf2f((bf16_t)g[j]) + ge[j];
      if (drop.seed_ptr) gy = kp[j] ? gy * drop.inv_keep : 0.f;
      const float h0 = bf2f((bf16_t)a[j]), h1 = bf2f((bf16_t)b[j]);
      d0[j] = gy * h1 * gelu_erf_grad(h0);
      d1[j] = gy * gelu_erf(h0);
    }
    *reinterpret_cast<uint4*>(dh + (long long)m * lddh + c) = make_uint4(pack2bf(d0[0], d0[1]), pack2bf(d0[2], d0[3]), pack2bf(d0[4], d0[5]), pack2bf(d0[6], d0[7]));
    *reinterpret_cast<uint4*>(dh + (long long)m * lddh + Nh + c) = make_uint4(pack2bf(d1[0], d1[1]), pack2bf(d1[2], d1[3]), pack2bf(d1[4], d1[5]), pack2bf(d1[6], d1[7]));
  }
}

// cross entropy over fp32 logits [R,V], ignore_index = -100, mean over valid rows (modeling_t5.py:1873-1877).
// loss += -(log softmax)[label] * inv_count ; dlogits (bf16) = (softmax - onehot) * inv_count * loss_scale.
// Round 4: the row terms are added in ROW order by the block that finishes last (ticket), not by fp32 atomics in arrival order: the loss of a
// step is the same bits on every run and on every rank of a replicated T5 (tests/test_frame_shard_gpu.py compares ranks bit for bit; with
// atomicAdd the sum of the 8-14 row terms changed in the last bit from run to run).  The terms travel through a library-owned scratch
// (write-through stores, one agent-scope acquire by the last arriver: cdna_hip_programming.md Guideline 16).  Round 6: the scratch is the
// caller's (mrblip_set_reduce_workspace; common.h) — launches that share one must be stream-ordered; without one, or with R > CE_MAX_ROWS,
// the sum falls back to atomics.
#define CE_MAX_ROWS MRB_RWS_CE_ROWS
__global__ __launch_bounds__(1024) void ce_kernel(const float* __restrict__ logits, long long ldl, const int* __restrict__ labels, int V,
                                                  float inv_count, float* loss, bf16_t* dlogits, long long ldd, float* g_ce_terms, unsigned int* g_ce_ticket,
                                                  const int* __restrict__ n_valid_dev) {
  const bool ordered = g_ce_terms != nullptr;
  // the mean's 1 / count from a DEVICE word (mrblip_cross_entropy_nvalid): a captured graph then serves batches with any number of valid
  // label tokens (correctly rounded fp32 division: the same bits as the host's float32(1) / float32(count))
  if (n_valid_dev) inv_count = 1.0f / (float)max(*n_valid_dev, 1);
  __shared__ float red[16];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* lr = logits + (long long)r * ldl;
  const int label = labels[r];
  float mx = NEG_INF_F;
  for (int i = threadIdx.x; i < V; i += 1024) mx = fmaxf(mx, lr[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 1024) s += __expf(lr[i] - mx);
  s = wave_sum(s);
  if (lane == 0) red[wv] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += red[i];
  const float lse = mx + __logf(s);
  const bool valid = label >= 0;
  if (threadIdx.x == 0) {
    const float term = valid ? (lse - lr[label]) * inv_count : 0.f;
    if (!ordered) {
      if (valid) atomicAdd(loss, term);
    } else {
      __hip_atomic_store(&g_ce_terms[r], term, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // write-through
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int old = __hip_atomic_fetch_add(g_ce_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == gridDim.x - 1) {
        __hip_atomic_store(g_ce_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        float acc = *loss;
        for (int i = 0; i < (int)gridDim.x; ++i) acc += __hip_atomic_load(&g_ce_terms[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *loss = acc;
      }
    }
  }
  if (dlogits) {
    bf16_t* dr = dlogits + (long long)r * ldd;
    for (int i = threadIdx.x; i < V; i += 1024) {
      float g = 0.f;
      if (valid) g = (__expf(lr[i] - lse) - (i == label ? 1.f : 0.f)) * inv_count;
      dr[i] = f2bf(g);
    }
  }
}

// AdamW on a flat fp32 segment (torch.optim.AdamW semantics; runner_base.py:102-132).  hyper = {lr, 1/bc1, 1/sqrt(bc2), grad_scale}
// lives in device memory so a captured graph replays with fresh values.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long long n, const float* __restrict__ hyper, float beta1, float beta2, float eps, float wd,
                                                    const uint32_t* __restrict__ guard) {
  // guard (optional): a device word that must be ZERO for the step to be applied — the error word of the in-GEMM thin role (gemm.hip):
  // a step whose activations / gradients may be wrong because a bounded wait ran out is dropped on the device, before the host knows
  if (guard && *guard != 0u) return;
  const float lr = hyper[0], ibc1 = hyper[1], isbc2 = hyper[2], gs = hyper[3];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i] * gs;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * isbc2 + eps;
    pi -= lr * ibc1 * (mi / denom);
    p[i] = pi;
  }
}

__global__ void seed_bump_kernel(uint32_t* seed) { if (threadIdx.x == 0) *seed = *seed * 1664525u + 1013904223u; }

// ---------------------------------------------------------------------------------------------------------
// LoRA (peft 0.13.0 Linear, r = 8):  y = W x + scale * B (A dropout(x)).   blip2_mr.py:182-200, 236.
// lora_dx_add: dx[m,k] += drop_mask(m,k) * sum_{r<R} G[m,r] * Acat[r,k]     (dx fp32 or bf16; G bf16 [M,ldg]; Acat bf16 [R,K], R <= 32)
template <bool DX_F32>
__global__ __launch_bounds__(256) void lora_dx_add_kernel(void* __restrict__ dx_, long long lddx, const bf16_t* __restrict__ G, long long ldg,
                                                          const bf16_t* __restrict__ A, int R, int M, int K, DropoutArg drop) {
  const long long total4 = (long long)M * (K / 4);
  const uint32_t seed = drop.seed_ptr ? *drop.seed_ptr : 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
    float add[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < R; r0 += 8) {
      const bf16x8 gv = *reinterpret_cast<const bf16x8*>(G + (long long)m * ldg + r0);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float gr = bf2f((bf16_t)gv[r]);
        const uint2 a = *reinterpret_cast<const uint2*>(A + (long long)(r0 + r) * K + k);
        add[0] += gr * bf2f((bf16_t)(a.x & 0xffff)); add[1] += gr * bf2f((bf16_t)(a.x >> 16));
        add[2] += gr * bf2f((bf16_t)(a.y & 0xffff)); add[3] += gr * bf2f((bf16_t)(a.y >> 16));
      }
    }
    if (drop.seed_ptr) {
      bool kq0, kq1, kq2, kq3;   // (k % 4 == 0, K % 4 == 0: two pair hashes)
      mrb_keep4((uint32_t)m * (uint32_t)K + (uint32_t)k, seed, drop.site, drop.thresh24, kq0, kq1, kq2, kq3);
      add[0] = kq0 ? add[0] * drop.inv_keep : 0.f; add[1] = kq1 ? add[1] * drop.inv_keep : 0.f;
      add[2] = kq2 ? add[2] * drop.inv_keep : 0.f; add[3] = kq3 ? add[3] * drop.inv_keep : 0.f;
    }
    if (DX_F32) {
      float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(dx_) + (long long)m * lddx + k);
      float4 v = *p;
      v.x += add[0]; v.y += add[1]; v.z += add[2]; v.w += add[3];
      *p = v;
    } else {
      uint2* p = reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(dx_) + (long long)m * lddx + k);
      const uint2 v = *p;
      const float x0 = bf2f((bf16_t)(v.x & 0xffff)) + add[0], x1 = bf2f((bf16_t)(v.x >> 16)) + add[1];
      const float x2 = bf2f((bf16_t)(v.y & 0xffff)) + add[2], x3 = bf2f((bf16_t)(v.y >> 16)) + add[3];
      *p = make_uint2(pack2bf(x0, x1), pack2bf(x2, x3));
    }
  }
}

// bf16 -> dropout -> bf16 (LoRA input dropout, peft lora_dropout): idx = row * N + col
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16_t* __restrict__ x, long long ldx, bf16_t* __restrict__ out, long long ldo, int M, int N, DropoutArg drop) {
  const long long total8 = (long long)M * (N / 8);
  const uint32_t seed = drop.seed_ptr ? *drop.seed_ptr : 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (N / 8)), c = (int)(i % (N / 8)) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (long long)m * ldx + c);
    float o[8];
    bool kq[8] = {true, true, true, true, true, true, true, true};
    if (drop.seed_ptr) {   // (c % 8 == 0, N % 8 == 0: element pairs share a hash)
#pragma unroll
      for (int j = 0; j < 8; j += 2) mrb_keep2((uint32_t)m * (uint32_t)N + (uint32_t)(c + j), seed, drop.site, drop.thresh24, kq[j], kq[j + 1]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = bf2f((bf16_t)v[j]);
      if (drop.seed_ptr) o[j] = kq[j] ? o[j] * drop.inv_keep : 0.f;
    }
    *reinterpret_cast<uint4*>(out + (long long)m * ldo + c) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
  }
}

// out[c] += sum_m x[m,c]  (bias gradients).  Round 4: the row blocks' partial sums are added in BLOCK order by the last-arriving block of a
// column group (ticket; write-through partials in a library-owned scratch, as ce_kernel) instead of fp32 atomics in arrival order: the
// t5_proj bias gradient is the same bits on every run.  Launches are expected to be stream-ordered.
#define CS_MAXY MRB_RWS_CS_MAXY
#define CS_MAXN MRB_RWS_CS_MAXN
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long long ldx, int M, int N, int rows_per_block, float* __restrict__ out,
                                                     float* g_cs_part, unsigned int* g_cs_ticket) {
  const bool ordered = g_cs_part != nullptr;
  __shared__ int last_flag;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int mbeg = blockIdx.y * rows_per_block, mend = min(M, mbeg + rows_per_block);
  float acc = 0.f;
  if (c < N)
    for (int m = mbeg; m < mend; ++m) acc += x[(long long)m * ldx + c];
  if (!ordered) {
    if (c < N) atomicAdd(out + c, acc);
    return;
  }
  if (c < N) __hip_atomic_store(&g_cs_part[blockIdx.y * CS_MAXN + c], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(&g_cs_ticket[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = old == gridDim.y - 1;
    if (last_flag) __hip_atomic_store(&g_cs_ticket[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last_flag && c < N) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float tot = out[c];
    for (int y = 0; y < (int)gridDim.y; ++y) tot += __hip_atomic_load(&g_cs_part[y * CS_MAXN + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[c] = tot;
  }
}

// LoRA master weights (fp32 flat buffer: A [8,K], Bt [8,out] per adapter) -> bf16 GEMM operands, one launch for all
// adapters via a device descriptor table of 10 int64 per adapter {a_off, bt_off, K, out, acat_off, wext_off, bblk_off, Ntot, acatt_off, 0}:
//   acat[acat_off + r*K + k]      = bf16(scale * A[r,k])        (stacked [8*nad, K] "down" operand of a group)
//   acatt[acatt_off + k*64 + r]   = bf16(scale * A[r,k])        ([K, 64] transposed copy: K-extension operand of the dX GEMM)
//   wext[wext_off + n*64 + r]     = bf16(Bt[r,n])               (K-extension operand of the forward GEMM)
//   bblk[bblk_off + r*Ntot + n]   = bf16(scale * Bt[r,n])       (block-diagonal [8*nad, Ntot] operand of g = dy @ B)
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* __restrict__ flat, bf16_t* __restrict__ acat, bf16_t* __restrict__ wext,
                                                        bf16_t* __restrict__ bblk, bf16_t* __restrict__ acatt, const long long* __restrict__ desc, float scale) {
  const long long* d = desc + (long long)blockIdx.y * 10;
  const long long a_off = d[0], bt_off = d[1], K = d[2], out = d[3], acat_off = d[4], wext_off = d[5], bblk_off = d[6], ntot = d[7], acatt_off = d[8];
  const long long stride = (long long)gridDim.x * 256, t0 = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long i = t0; i < 8 * K; i += stride) acat[acat_off + i] = f2bf(flat[a_off + i] * scale);
  for (long long k = t0; k < K; k += stride) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = flat[a_off + r * K + k] * scale;
    *reinterpret_cast<uint4*>(acatt + acatt_off + k * 64) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
  for (long long n = t0; n < out; n += stride) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = flat[bt_off + r * out + n];
    *reinterpret_cast<uint4*>(wext + wext_off + n * 64) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#pragma unroll
    for (int r = 0; r < 8; ++r) bblk[bblk_off + r * ntot + n] = f2bf(v[r] * scale);
  }
}

// ========================================================================================================= C ABI
static DropoutArg mk_drop(const uint32_t* seed_ptr, uint32_t site, float p) {
  DropoutArg d;
  d.seed_ptr = (p > 0.f) ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p);
  return d;
}
static int grid_for(long long work_items) { return (int)((work_items + 255) / 256 > 4096 ? 4096 : (work_items + 255) / 256 < 1 ? 1 : (work_items + 255) / 256); }

template <bool U8, bool F16>
static int patchify_launch(const void* video, const float* mean3, const float* std3, void* out16, int F, int IMG, int P, int Kpad, hipStream_t stream, const char* what) {
  MRB_REQUIRE(F > 0 && P > 0 && IMG % P == 0 && IMG % 4 == 0 && Kpad >= 3 * P * P && Kpad % 64 == 0 && (!U8 || (mean3 && std3)), "%s: bad arguments", what);
  MRB_REQUIRE(((uintptr_t)video % 16) == 0 && ((uintptr_t)out16 % 16) == 0, "%s: 16-B alignment", what);
  const int G = IMG / P;
  const int lds = G * Kpad * 2;
  MRB_REQUIRE(lds <= 64 * 1024, "%s: a patch row of %d x %d 16-bit values does not fit the staging tile", what, G, Kpad);
  const float m0 = U8 ? mean3[0] : 0.f, m1 = U8 ? mean3[1] : 0.f, m2 = U8 ? mean3[2] : 0.f, s0 = U8 ? std3[0] : 1.f, s1 = U8 ? std3[1] : 1.f, s2 = U8 ? std3[2] : 1.f;
  if (P == 14) hipLaunchKernelGGL((patchify_kernel<U8, F16, 14>), dim3(G, F), dim3(256), lds, stream, video, (bf16_t*)out16, IMG, P, G, Kpad, m0, m1, m2, s0, s1, s2);
  else hipLaunchKernelGGL((patchify_kernel<U8, F16, 0>), dim3(G, F), dim3(256), lds, stream, video, (bf16_t*)out16, IMG, P, G, Kpad, m0, m1, m2, s0, s1, s2);
  return mrblip_check_launch(what);
}
extern "C" int mrblip_patchify(const float* video, void* out_bf16, int F, int IMG, int P, int Kpad, hipStream_t stream) {
  return patchify_launch<false, false>(video, nullptr, nullptr, out_bf16, F, IMG, P, Kpad, stream, "patchify");
}
extern "C" int mrblip_patchify_u8(const uint8_t* video, const float* mean3, const float* std3, void* out_bf16, int F, int IMG, int P, int Kpad,
                                  hipStream_t stream) {
  return patchify_launch<true, false>(video, mean3, std3, out_bf16, F, IMG, P, Kpad, stream, "patchify_u8");
}
// fp16 patch rows (fp16-operand ViT, round 4)
extern "C" int mrblip_patchify_f16(const float* video, void* out_f16, int F, int IMG, int P, int Kpad, hipStream_t stream) {
  return patchify_launch<false, true>(video, nullptr, nullptr, out_f16, F, IMG, P, Kpad, stream, "patchify_f16");
}
extern "C" int mrblip_patchify_u8_f16(const uint8_t* video, const float* mean3, const float* std3, void* out_f16, int F, int IMG, int P, int Kpad,
                                      hipStream_t stream) {
  return patchify_launch<true, true>(video, mean3, std3, out_f16, F, IMG, P, Kpad, stream, "patchify_u8_f16");
}
extern "C" int mrblip_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int F, int NP, int D, hipStream_t stream) {
  MRB_REQUIRE(F > 0 && NP > 0 && D % 4 == 0, "vit_assemble: bad shape");
  hipLaunchKernelGGL(vit_assemble_kernel, dim3(NP + 1, F), dim3(256), 0, stream, patch, cls, pos, x, NP, D);
  return mrblip_check_launch("vit_assemble");
}
extern "C" int mrblip_row_copy(const float* src, long long lds_, const int* src_idx, float* dst, long long ldd, const int* dst_idx, int n_rows,
                               int D, int accumulate, hipStream_t stream) {
  MRB_REQUIRE(n_rows >= 0 && D % 4 == 0, "row_copy: bad shape");
  if (n_rows == 0) return MRBLIP_OK;
  if (accumulate) hipLaunchKernelGGL(row_copy_kernel<true>, dim3(n_rows), dim3(256), 0, stream, src, lds_, src_idx, dst, ldd, dst_idx, D);
  else hipLaunchKernelGGL(row_copy_kernel<false>, dim3(n_rows), dim3(256), 0, stream, src, lds_, src_idx, dst, ldd, dst_idx, D);
  return mrblip_check_launch("row_copy");
}
extern "C" int mrblip_mean_pool(const float* x, float* out, int F, int n, int D, hipStream_t stream) {
  MRB_REQUIRE(F > 0 && n > 0 && D % 4 == 0, "mean_pool: bad shape");
  hipLaunchKernelGGL(mean_pool_kernel, dim3((D / 4 + 31) / 32, F), dim3(64), 0, stream, x, out, n, D);
  return mrblip_check_launch("mean_pool");
}
extern "C" int mrblip_mean_pool_bwd(const float* dout, float* dx, int F, int n, int D, hipStream_t stream) {
  MRB_REQUIRE(F > 0 && n > 0 && D % 4 == 0, "mean_pool_bwd: bad shape");
  hipLaunchKernelGGL(mean_pool_bwd_kernel, dim3(n, F), dim3(256), 0, stream, dout, dx, n, D);
  return mrblip_check_launch("mean_pool_bwd");
}
extern "C" int mrblip_cast_dropout(const float* x, long long ldx, void* out_bf16, long long ldob, float* out_f32, long long ldof, int M, int N,
                                   const uint32_t* seed_ptr, uint32_t site, float p, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && N > 0 && N % 4 == 0, "cast_dropout: bad shape");
  hipLaunchKernelGGL(cast_drop_kernel, dim3(grid_for((long long)M * N / 4)), dim3(256), 0, stream, x, ldx, (bf16_t*)out_bf16, ldob, out_f32, ldof, M, N, mk_drop(seed_ptr, site, p));
  return mrblip_check_launch("cast_dropout");
}
extern "C" int mrblip_gelu_bwd(const void* dy, const void* h, void* dh, long long n, hipStream_t stream) {
  MRB_REQUIRE(n > 0 && n % 8 == 0, "gelu_bwd: n %% 8 != 0");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)h, (bf16_t*)dh, n / 8);
  return mrblip_check_launch("gelu_bwd");
}
extern "C" int mrblip_gated_gelu_bwd(const void* dy, long long lddy, const void* h, long long ldh, void* dh, long long lddh, int M, int Nh,
                                     const uint32_t* seed_ptr, uint32_t site, float p, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && Nh > 0 && Nh % 8 == 0, "gated_gelu_bwd: bad shape");
  hipLaunchKernelGGL(gated_bwd_kernel, dim3(grid_for((long long)M * Nh / 8)), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const bf16_t*)h, ldh, (bf16_t*)dh, lddh, M, Nh, mk_drop(seed_ptr, site, p));
  return mrblip_check_launch("gated_gelu_bwd");
}
// out = (residual) + part 0 + part 1 + ... (fp32, in part order): the K-split partial products of mrblip_gemm_ksplit for a consumer that has no
// parts form of its own (the encoder-output gradient that the decoder's stacked cross K / V projections accumulate chunk by chunk)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ parts, long long ldp, long long pstride, int nparts, const float* residual,
                                                        long long ldr, float* out, long long ldo, int M, int N) {
  const long long total4 = (long long)M * (N / 4);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (N / 4)), c = (int)(i % (N / 4)) * 4;
    float4 acc = residual ? *reinterpret_cast<const float4*>(residual + (long long)m * ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s_ = 0; s_ < nparts; ++s_) {
      const float4 t = *reinterpret_cast<const float4*>(parts + (long long)s_ * pstride + (long long)m * ldp + c);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(out + (long long)m * ldo + c) = acc;
  }
}
extern "C" int mrblip_sum_parts(const float* parts, long long ldp, long long pstride, int nparts, const float* residual, long long ldr, float* out,
                                long long ldo, int M, int N, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && N > 0 && (N % 4) == 0 && nparts >= 1 && (ldp % 4) == 0 && (pstride % 4) == 0 && (ldo % 4) == 0 && (!residual || (ldr % 4) == 0) &&
                  ((uintptr_t)parts % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)residual % 16) == 0, "sum_parts: 16-B aligned fp32 rows");
  hipLaunchKernelGGL(sum_parts_kernel, dim3(grid_for((long long)M * N / 4)), dim3(256), 0, stream, parts, ldp, pstride, nparts, residual, ldr, out, ldo, M, N);
  return mrblip_check_launch("sum_parts");
}
// ... with dy in two bf16 parts: dy + mask(ext_site, ext_p) (.) dy_ext (same leading dimension; dy_ext == nullptr: plain)
extern "C" int mrblip_gated_gelu_bwd_parts(const void* dy, const void* dy_ext, long long lddy, const void* h, long long ldh, void* dh, long long lddh, int M, int Nh,
                                           const uint32_t* seed_ptr, uint32_t site, float p, uint32_t ext_site, float ext_p, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && Nh > 0 && Nh % 8 == 0 && (lddy % 8) == 0 && ((uintptr_t)dy_ext % 16) == 0, "gated_gelu_bwd_parts: bad shape");
  MRB_REQUIRE(!(p > 0.f || ext_p > 0.f) || seed_ptr, "gated_gelu_bwd_parts: dropout needs a device seed pointer");
  hipLaunchKernelGGL(gated_bwd_kernel, dim3(grid_for((long long)M * Nh / 8)), dim3(256), 0, stream, (const bf16_t*)dy, lddy, (const bf16_t*)h, ldh, (bf16_t*)dh, lddh, M, Nh,
                     mk_drop(seed_ptr, site, p), (const bf16_t*)dy_ext, mk_drop(seed_ptr, ext_site, ext_p));
  return mrblip_check_launch("gated_gelu_bwd_parts");
}
extern "C" int mrblip_cross_entropy(const float* logits, long long ldl, const int* labels, int R, int V, float inv_count, float* loss,
                                    void* dlogits_bf16, long long ldd, hipStream_t stream) {
  MRB_REQUIRE(R > 0 && V > 0, "cross_entropy: bad shape");
  char* ws = R <= CE_MAX_ROWS ? mrblip_reduce_workspace() : nullptr;
  hipLaunchKernelGGL(ce_kernel, dim3(R), dim3(1024), 0, stream, logits, ldl, labels, V, inv_count, loss, (bf16_t*)dlogits_bf16, ldd,
                     ws ? (float*)(ws + MRB_RWS_OFF_CE) : nullptr, ws ? (unsigned int*)ws + MRB_RWS_TICKET_CE : nullptr, (const int*)nullptr);
  return mrblip_check_launch("cross_entropy");
}
// the same with the number of valid (label != -100) rows read from DEVICE memory: inv_count = 1 / max(*n_valid, 1)
extern "C" int mrblip_cross_entropy_nvalid(const float* logits, long long ldl, const int* labels, int R, int V, const int* n_valid, float* loss,
                                           void* dlogits_bf16, long long ldd, hipStream_t stream) {
  MRB_REQUIRE(R > 0 && V > 0 && n_valid, "cross_entropy_nvalid: bad shape / no count");
  char* ws = R <= CE_MAX_ROWS ? mrblip_reduce_workspace() : nullptr;
  hipLaunchKernelGGL(ce_kernel, dim3(R), dim3(1024), 0, stream, logits, ldl, labels, V, 0.f, loss, (bf16_t*)dlogits_bf16, ldd,
                     ws ? (float*)(ws + MRB_RWS_OFF_CE) : nullptr, ws ? (unsigned int*)ws + MRB_RWS_TICKET_CE : nullptr, n_valid);
  return mrblip_check_launch("cross_entropy_nvalid");
}
extern "C" int mrblip_adamw(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2, float eps,
                            float weight_decay, hipStream_t stream) {
  if (n <= 0) return MRBLIP_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, n, hyper, beta1, beta2, eps, weight_decay, (const uint32_t*)nullptr);
  return mrblip_check_launch("adamw");
}
extern "C" int mrblip_adamw_guarded(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2, float eps,
                                    float weight_decay, const uint32_t* guard, hipStream_t stream) {
  MRB_REQUIRE(p && g && m && v && hyper && n >= 0, "adamw_guarded: null operand");
  if (n == 0) return MRBLIP_OK;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, n, hyper, beta1, beta2, eps, weight_decay, guard);
  return mrblip_check_launch("adamw_guarded");
}
// Pull a byte range through the memory-side cache ahead of its consumer (round 4): the weights of a T5 layer are touched once per pass,
// so every GEMM finds them in HBM — stand-alone loops over one weight set (which the 256 MB Infinity Cache holds) are 10-25 % faster
// than the same launches in the step.  A few blocks on a side stream read the NEXT launch's operands while the current one computes;
// the loads are dropped (the xor below can never match), nothing is written.
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, long long n16, uint32_t* sink) {
  uint32_t acc = 0;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long step = (long long)gridDim.x * 256;
  for (; i + 3 * step < n16; i += 4 * step) {
    const uint4 a = p[i], b = p[i + step], c = p[i + 2 * step], d = p[i + 3 * step];
    acc |= (a.x ^ b.x ^ c.x ^ d.x) & (a.y ^ b.y ^ c.y ^ d.y) & (a.z ^ b.z ^ c.z ^ d.z) & (a.w ^ b.w ^ c.w ^ d.w);
  }
  for (; i < n16; i += step) { const uint4 a = p[i]; acc |= a.x & a.y & a.z & a.w; }
  if (acc == 0x9e3779b9u && sink) *sink = acc;   // keeps the loads alive; never true for bf16 weight data in practice, harmless if it is
}
extern "C" int mrblip_prefetch(const void* ptr, long long bytes, int n_blocks, hipStream_t stream) {
  MRB_REQUIRE(ptr && bytes >= 0 && ((uintptr_t)ptr % 16) == 0 && n_blocks > 0 && n_blocks <= 1024, "prefetch: 16-byte aligned range, 1..1024 blocks");
  if (bytes < 16) return MRBLIP_OK;
  static uint32_t* sink = nullptr;
  if (!sink && hipMalloc((void**)&sink, 64) != hipSuccess) { mrblip_set_error("prefetch: cannot allocate the sink word"); return MRBLIP_ELAUNCH; }
  hipLaunchKernelGGL(prefetch_kernel, dim3(n_blocks), dim3(256), 0, stream, (const uint4*)ptr, bytes / 16, sink);
  return mrblip_check_launch("prefetch");
}
extern "C" int mrblip_seed_bump(uint32_t* seed, hipStream_t stream) {
  hipLaunchKernelGGL(seed_bump_kernel, dim3(1), dim3(64), 0, stream, seed);
  return mrblip_check_launch("seed_bump");
}
extern "C" int mrblip_lora_dx_add(void* dx, long long lddx, int dx_f32, const void* G, long long ldg, const void* Acat_bf16, int R, int M, int K,
                                  const uint32_t* seed_ptr, uint32_t site, float p, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && ldg % 8 == 0 && R > 0 && R <= 32 && R % 8 == 0, "lora_dx_add: bad shape");
  if (dx_f32) hipLaunchKernelGGL(lora_dx_add_kernel<true>, dim3(grid_for((long long)M * K / 4)), dim3(256), 0, stream, dx, lddx, (const bf16_t*)G, ldg, (const bf16_t*)Acat_bf16, R, M, K, mk_drop(seed_ptr, site, p));
  else hipLaunchKernelGGL(lora_dx_add_kernel<false>, dim3(grid_for((long long)M * K / 4)), dim3(256), 0, stream, dx, lddx, (const bf16_t*)G, ldg, (const bf16_t*)Acat_bf16, R, M, K, mk_drop(seed_ptr, site, p));
  return mrblip_check_launch("lora_dx_add");
}
// the same for `groups` LoRA groups that share the input (round 4): dx[m, k] += sum_g mask_g[m, k] * (G[m, g * g_gstride : + R] . A[g * a_gstride ...][:, k]),
// mask_g drawn with call-site id site0 + g * site_stride — the rank-R parts of the input gradient of ALL decoder layers' cross-attention
// K / V adapters (each layer masked its own copy of the encoder output) in one pass over dx instead of one read-modify-write per layer.
__global__ __launch_bounds__(256) void lora_dx_add_batched_kernel(float* __restrict__ dx, long long lddx, const bf16_t* __restrict__ G, long long ldg,
                                                                  long long g_gstride, const bf16_t* __restrict__ A, long long a_gstride, int R, int M,
                                                                  int K, int groups, uint32_t site_stride, DropoutArg drop) {
  const long long total4 = (long long)M * (K / 4);
  const uint32_t seed = drop.seed_ptr ? mrb_seed_load(drop.seed_ptr) : 0u;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < groups; ++g) {
      float add[4] = {0.f, 0.f, 0.f, 0.f};
      const bf16_t* Ag = A + g * a_gstride;
      for (int r0 = 0; r0 < R; r0 += 8) {
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(G + (long long)m * ldg + g * g_gstride + r0);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float gr = bf2f((bf16_t)gv[r]);
          const uint2 a = *reinterpret_cast<const uint2*>(Ag + (long long)(r0 + r) * K + k);
          add[0] += gr * bf2f((bf16_t)(a.x & 0xffff)); add[1] += gr * bf2f((bf16_t)(a.x >> 16));
          add[2] += gr * bf2f((bf16_t)(a.y & 0xffff)); add[3] += gr * bf2f((bf16_t)(a.y >> 16));
        }
      }
      if (drop.seed_ptr) {
        const uint32_t site = drop.site + (uint32_t)g * site_stride;
        bool k0, k1, k2, k3;
        mrb_keep2((uint32_t)m * (uint32_t)K + (uint32_t)k, seed, site, drop.thresh24, k0, k1);
        mrb_keep2((uint32_t)m * (uint32_t)K + (uint32_t)k + 2u, seed, site, drop.thresh24, k2, k3);
        tot[0] += k0 ? add[0] * drop.inv_keep : 0.f; tot[1] += k1 ? add[1] * drop.inv_keep : 0.f;
        tot[2] += k2 ? add[2] * drop.inv_keep : 0.f; tot[3] += k3 ? add[3] * drop.inv_keep : 0.f;
      } else {
        tot[0] += add[0]; tot[1] += add[1]; tot[2] += add[2]; tot[3] += add[3];
      }
    }
    float4* q = reinterpret_cast<float4*>(dx + (long long)m * lddx + k);
    float4 v = *q;
    v.x += tot[0]; v.y += tot[1]; v.z += tot[2]; v.w += tot[3];
    *q = v;
  }
}
extern "C" int mrblip_lora_dx_add_batched(float* dx, long long lddx, const void* G, long long ldg, long long g_gstride, const void* A, long long a_gstride,
                                          int R, int M, int K, int groups, const uint32_t* seed_ptr, uint32_t site0, uint32_t site_stride, float p,
                                          hipStream_t stream) {
  MRB_REQUIRE(M > 0 && K > 0 && K % 4 == 0 && ldg % 8 == 0 && g_gstride % 8 == 0 && a_gstride % 4 == 0 && lddx % 4 == 0 && R > 0 && R <= 32 && R % 8 == 0 && groups > 0,
              "lora_dx_add_batched: bad shape");
  hipLaunchKernelGGL(lora_dx_add_batched_kernel, dim3(grid_for((long long)M * K / 4)), dim3(256), 0, stream, dx, lddx, (const bf16_t*)G, ldg, g_gstride,
                     (const bf16_t*)A, a_gstride, R, M, K, groups, site_stride, mk_drop(seed_ptr, site0, p));
  return mrblip_check_launch("lora_dx_add_batched");
}
extern "C" int mrblip_dropout_bf16(const void* x, long long ldx, void* out, long long ldo, int M, int N, const uint32_t* seed_ptr, uint32_t site,
                                   float p, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && N > 0 && N % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0, "dropout_bf16: bad shape");
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid_for((long long)M * N / 8)), dim3(256), 0, stream, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, M, N, mk_drop(seed_ptr, site, p));
  return mrblip_check_launch("dropout_bf16");
}
extern "C" int mrblip_colsum(const float* x, long long ldx, int M, int N, float* out, hipStream_t stream) {
  MRB_REQUIRE(M > 0 && N > 0, "colsum: bad shape");
  int rpb = 64;
  char* ws = N <= CS_MAXN ? mrblip_reduce_workspace() : nullptr;
  const int ordered = ws ? 1 : 0;
  if (ordered && (M + rpb - 1) / rpb > CS_MAXY) rpb = (M + CS_MAXY - 1) / CS_MAXY;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, (M + rpb - 1) / rpb), dim3(256), 0, stream, x, ldx, M, N, rpb, out,
                     ws ? (float*)(ws + MRB_RWS_OFF_CS) : nullptr, ws ? (unsigned int*)ws + MRB_RWS_TICKET_CS : nullptr);
  return mrblip_check_launch("colsum");
}
extern "C" int mrblip_lora_pack(const float* flat, void* acat_bf16, void* wext_bf16, void* bblk_bf16, void* acatt_bf16, const long long* desc,
                                int n_adapters, float scale, hipStream_t stream) {
  MRB_REQUIRE(n_adapters > 0, "lora_pack: bad shape");
  hipLaunchKernelGGL(lora_pack_kernel, dim3(16, n_adapters), dim3(256), 0, stream, flat, (bf16_t*)acat_bf16, (bf16_t*)wext_bf16, (bf16_t*)bblk_bf16,
                     (bf16_t*)acatt_bf16, desc, scale);
  return mrblip_check_launch("lora_pack");
}
