// Fused LayerNorm / T5 RMSNorm, forward and backward, fp32 statistics (HBM-bound row kernels).
// One wave owns one row (D <= 2048, D % 4 == 0), the row lives in registers between the passes, so every
// row is read once and written once.
//
// Reference call sites: eva_vit.py:157,163 (norm1/norm2, eps 1e-6), blip2.py:113-119 (ln_vision, fp32),
// Qformer.py:104-107, 285-289, 372-375 (post-LN, eps 1e-12), modeling_t5.py:254-277 (T5LayerNorm = RMS).
#include "common.h"

#define NORM_MAXV 8  // float4 per lane -> D <= 64*4*8 = 2048

template <bool RMS, bool F16 = false>   // F16: the 16-bit output is IEEE fp16 (the frozen ViT's GEMM operands, round 4) instead of bf16
__global__ __launch_bounds__(256) void norm_fwd_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int M, int D, float eps, bf16_t* out_b,
                                                       long long ldob, float* out_f, long long ldof) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = D >> 2;
  // gamma / beta: the lane's columns are the same for every row it visits — fetched ONCE, before the row loop, all loads in flight
  // together.  (Round 3, ISA reading: loaded inside the per-column `if` of the store loop, each was a load + s_waitcnt vmcnt(0) pair —
  // up to 16 L2 round trips in series per row — and every column's block then also waited for the previous column's STORES, because
  // the compiler must assume loads pending at the join of the skipped branches: MRB_ALL_LOADS_DONE() tells it there are none.)
  float4 g4[NORM_MAXV], b4[NORM_MAXV];
#pragma unroll
  for (int j = 0; j < NORM_MAXV; ++j) {
    const int i = lane + 64 * j;
    const int ic = i < nv ? i : 0;   // clamped: unconditional load, value unused past the row's end
    g4[j] = reinterpret_cast<const float4*>(gamma)[ic];
    b4[j] = (!RMS && beta) ? reinterpret_cast<const float4*>(beta)[ic] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * ldx);
    float4 v[NORM_MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      v[j] = (i < nv) ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!RMS) s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    float mean = 0.f;
    if (!RMS) mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
        q += a * a + b * b + c * c + d * d;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    MRB_ALL_LOADS_DONE();
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        const float4 g = g4[j];
        float4 o;
        o.x = (v[j].x - mean) * rstd * g.x; o.y = (v[j].y - mean) * rstd * g.y;
        o.z = (v[j].z - mean) * rstd * g.z; o.w = (v[j].w - mean) * rstd * g.w;
        if (!RMS && beta) {
          const float4 b = b4[j];
          o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
        }
        if (out_f) reinterpret_cast<float4*>(out_f + (long long)row * ldof)[i] = o;
        if (out_b) reinterpret_cast<uint2*>(out_b + (long long)row * ldob)[i] = make_uint2(pack2x<F16>(o.x, o.y), pack2x<F16>(o.z, o.w));
      }
    }
  }
}

#define NORM_DW_MAXD MRB_RWS_DW_MAXD
#define NORM_DW_MAXB MRB_RWS_DW_MAXB
typedef uint32_t norm_u32x4 __attribute__((ext_vector_type(4)));
// ordered weight gradients: [block][dgamma | dbeta][D] partial sums + a ticket, both in the CALLER's reduce workspace (common.h; round 6)

// dx = dx_add + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma   (RMS: no mean(g) term, xhat = x*rstd)
// dgamma += sum_rows dy * xhat, dbeta += sum_rows dy  (optional, fp32 atomics, one per column per block)
// GOUT (round 6, mrblip_rmsnorm_bwd_parts_g): the launch also computes the LoRA "g" product of the operand it writes,
//   g_out[row, 0:8] = sum_c out_b[row, c] * g_b[r, c]      (g_b = scale * B^T of the NEXT projection's adapter, bf16 [8, D])
// — what a lora_thin launch over out_b would compute (19 + 10 us per T5 encoder layer of the backward, each a pass over a 8 MB operand that
// this kernel holds in registers): g_b is staged in LDS once per block (32 KB at D = 2048), the products run on v_dot2c_f32_bf16 over the
// PACKED bf16 pairs the store writes anyway (exact products, fp32 accumulation, as on the matrix cores), one wave reduction per r.
typedef __bf16 norm_bf2 __attribute__((ext_vector_type(2)));
template <bool RMS, bool GOUT = false>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                       long long ldx, const float* __restrict__ gamma, int M, int D, float eps,
                                                       const float* dx_add, long long ldadd, float* dx, long long lddx,
                                                       float* dgamma, float* dbeta, bf16_t* out_b = nullptr, long long ldob = 0,
                                                       DropoutArg drop = DropoutArg{nullptr, 0u, 0u, 1.0f}, float* g_dw_part = nullptr,
                                                       int nparts = 1, long long pstride = 0, int ext_part = 0,
                                                       DropoutArg ext_drop = DropoutArg{nullptr, 0u, 0u, 1.0f},
                                                       const bf16_t* __restrict__ g_b = nullptr, long long ldgb = 0, bf16_t* g_out = nullptr, long long ldg = 0) {
  // nparts > 1 (round 5, mrblip_rmsnorm_bwd_parts): dy arrives as PARTIAL products pstride elements apart (mrblip_gemm_ksplit) and is
  // their sum in part order; ext_part: the last part is the LoRA term g A, added under the lora_dropout keep mask of ext_drop over [M, D]
  __shared__ float red[2][4][64 * 4];
  __shared__ int last_flag;
  __shared__ uint2 gb_sh[GOUT ? 8 * (NORM_MAXV * 64) : 1];   // g_b as [r][column quad] (GOUT)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nv = D >> 2;
  if (GOUT) {
    for (int q = threadIdx.x; q < 8 * nv; q += 256) {
      const int r = q / nv, i = q - r * nv;
      gb_sh[r * (NORM_MAXV * 64) + i] = *reinterpret_cast<const uint2*>(g_b + (long long)r * ldgb + 4 * i);
    }
    __syncthreads();
  }
  float4 ag[NORM_MAXV], ab[NORM_MAXV];
  const bool want_dw = dgamma != nullptr;
  const bool ordered_dw = g_dw_part != nullptr;
  unsigned int* const g_dw_ticket = reinterpret_cast<unsigned int*>(g_dw_part) - (MRB_RWS_OFF_DW / 4) + MRB_RWS_TICKET_DW;   // (same workspace)
  const uint32_t seed = drop.seed_ptr ? mrb_seed_load(drop.seed_ptr) : ext_drop.seed_ptr ? mrb_seed_load(ext_drop.seed_ptr) : 0u;
  if (want_dw) {
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = blockIdx.x * 4 + wv; row < M; row += gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * ldx);
    const float4* dr = reinterpret_cast<const float4*>(dy + (long long)row * lddy);
    float4 v[NORM_MAXV], g[NORM_MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      v[j] = (i < nv) ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      g[j] = (i < nv) ? dr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int s_ = 1; s_ < nparts; ++s_) {
      const float4* ds = reinterpret_cast<const float4*>(dy + (long long)s_ * pstride + (long long)row * lddy);
      float4 t[NORM_MAXV];
#pragma unroll
      for (int j = 0; j < NORM_MAXV; ++j) {
        const int i = lane + 64 * j;
        t[j] = (i < nv) ? ds[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      const bool masked = ext_part && s_ == nparts - 1;
#pragma unroll
      for (int j = 0; j < NORM_MAXV; ++j) {
        if (masked) {
          bool k0 = true, k1 = true, k2 = true, k3 = true;
          if (ext_drop.seed_ptr) mrb_keep4((uint32_t)row * (uint32_t)D + (uint32_t)(4 * (lane + 64 * j)), seed, ext_drop.site, ext_drop.thresh24, k0, k1, k2, k3);
          t[j].x = k0 ? t[j].x * ext_drop.inv_keep : 0.f; t[j].y = k1 ? t[j].y * ext_drop.inv_keep : 0.f;
          t[j].z = k2 ? t[j].z * ext_drop.inv_keep : 0.f; t[j].w = k3 ? t[j].w * ext_drop.inv_keep : 0.f;
        }
        g[j].x += t[j].x; g[j].y += t[j].y; g[j].z += t[j].z; g[j].w += t[j].w;
      }
    }
    // gamma and the dx_add rows ride with them (round 3, ISA reading: loaded inside the loops that use them, each of these — and the
    // dropout seed — was a load + s_waitcnt vmcnt(0) pair per column: ~24 round trips in series per row, the kernel's whole 24 us)
    float4 wq[NORM_MAXV], aq[NORM_MAXV];
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      const int ic = i < nv ? i : 0;
      wq[j] = reinterpret_cast<const float4*>(gamma)[ic];
      aq[j] = dx_add ? reinterpret_cast<const float4*>(dx_add + (long long)row * ldadd)[ic] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j)
      if (!RMS) s += v[j].x + v[j].y + v[j].z + v[j].w;
    float mean = 0.f;
    if (!RMS) mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        v[j].x *= rstd; v[j].y *= rstd; v[j].z *= rstd; v[j].w *= rstd;  // xhat
        if (want_dw) {
          ag[j].x += g[j].x * v[j].x; ag[j].y += g[j].y * v[j].y; ag[j].z += g[j].z * v[j].z; ag[j].w += g[j].w * v[j].w;
          ab[j].x += g[j].x; ab[j].y += g[j].y; ab[j].z += g[j].z; ab[j].w += g[j].w;
        }
        const float4 w = wq[j];
        g[j].x *= w.x; g[j].y *= w.y; g[j].z *= w.z; g[j].w *= w.w;
        sg += g[j].x + g[j].y + g[j].z + g[j].w;
        sgx += g[j].x * v[j].x + g[j].y * v[j].y + g[j].z * v[j].z + g[j].w * v[j].w;
      }
    }
    const float mg = RMS ? 0.f : wave_sum(sg) / (float)D;
    const float mgx = wave_sum(sgx) / (float)D;
    MRB_ALL_LOADS_DONE();
    float gacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      const int i = lane + 64 * j;
      if (i < nv) {
        float4 o;
        o.x = rstd * (g[j].x - mg - v[j].x * mgx); o.y = rstd * (g[j].y - mg - v[j].y * mgx);
        o.z = rstd * (g[j].z - mg - v[j].z * mgx); o.w = rstd * (g[j].w - mg - v[j].w * mgx);
        if (dx_add) {
          const float4 a = aq[j];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        if (dx) reinterpret_cast<float4*>(dx + (long long)row * lddx)[i] = o;
        if (out_b) {  // the consumer's bf16 GEMM operand = dropout-backward(dx) (what a separate mrblip_cast_dropout launch would write)
          if (drop.seed_ptr) {
            const uint32_t base = (uint32_t)row * (uint32_t)D + (uint32_t)(4 * i);
            bool k0, k1, k2, k3;   // (D % 4 == 0: base is even)
            mrb_keep4(base, seed, drop.site, drop.thresh24, k0, k1, k2, k3);
            o.x = k0 ? o.x * drop.inv_keep : 0.f;
            o.y = k1 ? o.y * drop.inv_keep : 0.f;
            o.z = k2 ? o.z * drop.inv_keep : 0.f;
            o.w = k3 ? o.w * drop.inv_keep : 0.f;
          }
          const uint2 pk = make_uint2(pack2bf(o.x, o.y), pack2bf(o.z, o.w));
          reinterpret_cast<uint2*>(out_b + (long long)row * ldob)[i] = pk;
          if (GOUT) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              const uint2 b = gb_sh[r * (NORM_MAXV * 64) + i];
              gacc[r] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(norm_bf2, pk.x), __builtin_bit_cast(norm_bf2, b.x), gacc[r], false);
              gacc[r] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(norm_bf2, pk.y), __builtin_bit_cast(norm_bf2, b.y), gacc[r], false);
            }
          }
        }
      }
    }
    if (GOUT) {
#pragma unroll
      for (int r = 0; r < 8; ++r) gacc[r] = wave_sum_uniform(gacc[r]);
      if (lane == 0)
        *reinterpret_cast<uint4*>(g_out + (long long)row * ldg) =
            make_uint4(pack2bf(gacc[0], gacc[1]), pack2bf(gacc[2], gacc[3]), pack2bf(gacc[4], gacc[5]), pack2bf(gacc[6], gacc[7]));
    }
  }
  if (want_dw) {
    // reduce the 4 waves of the block through LDS, then one atomic per column
#pragma unroll
    for (int j = 0; j < NORM_MAXV; ++j) {
      if (64 * j >= nv) break;  // uniform
      float* r0 = &red[0][wv][lane * 4];
      float* r1 = &red[1][wv][lane * 4];
      r0[0] = ag[j].x; r0[1] = ag[j].y; r0[2] = ag[j].z; r0[3] = ag[j].w;
      r1[0] = ab[j].x; r1[1] = ab[j].y; r1[2] = ab[j].z; r1[3] = ab[j].w;
      __syncthreads();
      if (wv == 0) {
        const int i = lane + 64 * j;
        if (i < nv) {
          float sgm[4], sbt[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            sgm[c] = red[0][0][lane * 4 + c] + red[0][1][lane * 4 + c] + red[0][2][lane * 4 + c] + red[0][3][lane * 4 + c];
            sbt[c] = red[1][0][lane * 4 + c] + red[1][1][lane * 4 + c] + red[1][2][lane * 4 + c] + red[1][3][lane * 4 + c];
          }
          if (ordered_dw) {   // this block's partial sums, write-through (see below)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g_dw_part + (long long)blockIdx.x * 2 * NORM_DW_MAXD, 0, 2 * NORM_DW_MAXD * 4, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(norm_u32x4, f32x4{sgm[0], sgm[1], sgm[2], sgm[3]}), rs, (uint32_t)(i * 16), 0, 16);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(norm_u32x4, f32x4{sbt[0], sbt[1], sbt[2], sbt[3]}), rs, (uint32_t)((NORM_DW_MAXD + i * 4) * 4), 0, 16);
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              atomicAdd(dgamma + i * 4 + c, sgm[c]);
              if (dbeta) atomicAdd(dbeta + i * 4 + c, sbt[c]);
            }
          }
        }
      }
      __syncthreads();
    }
    // Round 4: ORDERED weight gradients.  fp32 atomics add the blocks' partial sums in arrival order, so dgamma / dbeta (ln_vision: the
    // only trainable norm) changed in their last bits from run to run — the one thing left that made a train step irreproducible
    // (tools/determinism_check.py).  Each block now publishes its partial row [2][D] to a library-owned scratch with write-through stores,
    // drains them and takes a ticket; the block that draws the last ticket adds the partials in BLOCK order (cdna_hip_programming.md
    // Guideline 16, form R1).  Launches are expected to be stream-ordered (one scratch).
    if (ordered_dw) {
      if (wv == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(g_dw_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = old == gridDim.x - 1;
        if (last_flag) __hip_atomic_store(g_dw_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (last_flag) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int nb = (int)gridDim.x;
        for (int q = threadIdx.x; q < 2 * nv; q += 256) {          // float4 column group q of [dgamma | dbeta]
          const int arr = q >= nv ? 1 : 0, i = q - arr * nv;
          if (arr == 1 && !dbeta) continue;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const uint32_t off = (uint32_t)((arr * NORM_DW_MAXD + i * 4) * 4);
          for (int b0 = 0; b0 < nb; b0 += 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int b = min(b0 + u, nb - 1);
              const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g_dw_part + (long long)b * 2 * NORM_DW_MAXD, 0, 2 * NORM_DW_MAXD * 4, 0x00020000);
              t[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (b0 + u < nb) acc += t[u];
          }
          float* dst = (arr ? dbeta : dgamma) + i * 4;
          float4 cur = *reinterpret_cast<float4*>(dst);
          cur.x += acc[0]; cur.y += acc[1]; cur.z += acc[2]; cur.w += acc[3];
          *reinterpret_cast<float4*>(dst) = cur;
        }
      }
    }
  }
}

static int norm_check(int M, int D, const void* x, long long ldx) {
  MRB_REQUIRE(M > 0 && D > 0 && D <= 2048 && (D % 4) == 0, "norm: need 0 < D <= 2048, D %% 4 == 0 (M=%d D=%d)", M, D);
  MRB_REQUIRE(((uintptr_t)x % 16) == 0 && (ldx % 4) == 0, "norm: input must be 16-B aligned");
  return MRBLIP_OK;
}

extern "C" int mrblip_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, int M, int D, float eps,
                                    void* out_bf16, long long ldob, float* out_f32, long long ldof, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(out_bf16 || out_f32, "layernorm_fwd: no output");
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_fwd_kernel<false>, dim3(grid), dim3(256), 0, stream, x, ldx, gamma, beta, M, D, eps, (bf16_t*)out_bf16, ldob, out_f32, ldof);
  return mrblip_check_launch("layernorm_fwd");
}

// the same with an fp16 16-bit output (ViT norm1 / norm2 in fp16-operand mode; eva_vit.py:157-163 under fp16 autocast)
extern "C" int mrblip_layernorm_fwd_f16(const float* x, long long ldx, const float* gamma, const float* beta, int M, int D, float eps,
                                        void* out_f16, long long ldob, float* out_f32, long long ldof, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(out_f16 || out_f32, "layernorm_fwd_f16: no output");
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL((norm_fwd_kernel<false, true>), dim3(grid), dim3(256), 0, stream, x, ldx, gamma, beta, M, D, eps, (bf16_t*)out_f16, ldob, out_f32, ldof);
  return mrblip_check_launch("layernorm_fwd_f16");
}

extern "C" int mrblip_rmsnorm_fwd(const float* x, long long ldx, const float* weight, int M, int D, float eps, void* out_bf16,
                                  long long ldob, float* out_f32, long long ldof, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(out_bf16 || out_f32, "rmsnorm_fwd: no output");
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_fwd_kernel<true>, dim3(grid), dim3(256), 0, stream, x, ldx, weight, (const float*)nullptr, M, D, eps, (bf16_t*)out_bf16, ldob, out_f32, ldof);
  return mrblip_check_launch("rmsnorm_fwd");
}

extern "C" int mrblip_layernorm_bwd(const float* dy, long long lddy, const float* x, long long ldx, const float* gamma, int M, int D,
                                    float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, float* dgamma,
                                    float* dbeta, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  // (dgamma / dbeta: at most NORM_DW_MAXB blocks, whose partial sums the last one adds in block order; MRB_NORM_DW_ATOMIC=1: fp32 atomics)
  static int dw_atomic = -1;
  if (dw_atomic < 0) { const char* e = getenv("MRB_NORM_DW_ATOMIC"); dw_atomic = (e && e[0] == '1') ? 1 : 0; }
  char* ws = (dgamma && !dw_atomic && ((uintptr_t)dgamma % 16) == 0 && (!dbeta || ((uintptr_t)dbeta % 16) == 0)) ? mrblip_reduce_workspace() : nullptr;
  const int ordered = ws ? 1 : 0;
  const int grid = min((M + 3) / 4, dgamma ? (ordered ? NORM_DW_MAXB : 512) : 2048);
  hipLaunchKernelGGL(norm_bwd_kernel<false>, dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, gamma, M, D, eps, dx_add, ldadd, dx, lddx, dgamma, dbeta,
                     (bf16_t*)nullptr, 0ll, DropoutArg{nullptr, 0u, 0u, 1.0f}, ws ? (float*)(ws + MRB_RWS_OFF_DW) : nullptr);
  return mrblip_check_launch("layernorm_bwd");
}

// layernorm_bwd (dx only: frozen gamma / beta) that also writes the NEXT GEMM's operand: out_bf16 = bf16(dropout-backward(dx)) with the mask
// of (seed, site, p) over element index row * D + col — bit-identical to mrblip_cast_dropout(dx) in a second launch (round 6: the Q-Former's
// post-LayerNorm sub-layers, Qformer.py:285-289, 372-375 — 29 launches per step off the backward's dependent chain)
extern "C" int mrblip_layernorm_bwd_cast(const float* dy, long long lddy, const float* x, long long ldx, const float* gamma, int M, int D,
                                         float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, void* out_bf16,
                                         long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(out_bf16 && (ldob % 4) == 0, "layernorm_bwd_cast: bf16 output missing / unaligned");
  MRB_REQUIRE(!(p_drop > 0.f) || seed_ptr, "layernorm_bwd_cast: dropout needs a device seed pointer");
  DropoutArg d;
  d.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p_drop);
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_bwd_kernel<false>, dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, gamma, M, D, eps, dx_add, ldadd, dx, lddx, (float*)nullptr,
                     (float*)nullptr, (bf16_t*)out_bf16, ldob, d);
  return mrblip_check_launch("layernorm_bwd_cast");
}

// rmsnorm_bwd that also writes the NEXT GEMM's operand: out_bf16 = bf16(dropout-backward(dx)) with the mask of (seed, site, p) over
// element index row * D + col — bit-identical to mrblip_cast_dropout(dx) in a second launch (T5 sub-layer dropout, modeling_t5.py:613-615)
extern "C" int mrblip_rmsnorm_bwd_cast(const float* dy, long long lddy, const float* x, long long ldx, const float* weight, int M, int D,
                                       float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, void* out_bf16,
                                       long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(out_bf16 && (ldob % 4) == 0, "rmsnorm_bwd_cast: bf16 output missing / unaligned");
  DropoutArg d;
  d.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr;
  d.site = site;
  d.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f);
  d.inv_keep = 1.0f / (1.0f - p_drop);
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_bwd_kernel<true>, dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, weight, M, D, eps, dx_add, ldadd, dx, lddx, (float*)nullptr,
                     (float*)nullptr, (bf16_t*)out_bf16, ldob, d);
  return mrblip_check_launch("rmsnorm_bwd_cast");
}

// rmsnorm_bwd (+ the consumer's bf16 operand when out_bf16 != nullptr) with dy in nparts fp32 PARTS, pstride elements apart — the K-split
// input gradient of mrblip_gemm_ksplit; ext_part != 0: the last part is the LoRA term, added under the keep mask (ext_site, ext_p) over [M, D]
extern "C" int mrblip_rmsnorm_bwd_parts(const float* dy, long long lddy, int nparts, long long pstride, int ext_part, uint32_t ext_site, float ext_p,
                                        const float* x, long long ldx, const float* weight, int M, int D, float eps, const float* dx_add, long long ldadd,
                                        float* dx, long long lddx, void* out_bf16, long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                                        hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(nparts >= 1 && nparts <= 32 && (pstride % 4) == 0 && (lddy % 4) == 0 && ((uintptr_t)dy % 16) == 0, "rmsnorm_bwd_parts: 1..32 parts of 16-B aligned rows");
  MRB_REQUIRE(!out_bf16 || (ldob % 4) == 0, "rmsnorm_bwd_parts: bf16 output unaligned");
  MRB_REQUIRE(!(p_drop > 0.f || ext_p > 0.f) || seed_ptr, "rmsnorm_bwd_parts: dropout needs a device seed pointer");
  DropoutArg d, de;
  d.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr; d.site = site; d.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f); d.inv_keep = 1.0f / (1.0f - p_drop);
  de.seed_ptr = (ext_p > 0.f) ? seed_ptr : nullptr; de.site = ext_site; de.thresh24 = (uint32_t)(ext_p * 65536.0f + 0.5f); de.inv_keep = 1.0f / (1.0f - ext_p);
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_bwd_kernel<true>, dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, weight, M, D, eps, dx_add, ldadd, dx, lddx, (float*)nullptr,
                     (float*)nullptr, (bf16_t*)out_bf16, ldob, d, (float*)nullptr, nparts, pstride, ext_part, de);
  return mrblip_check_launch("rmsnorm_bwd_parts");
}

// mrblip_rmsnorm_bwd_parts that ALSO writes g_out[M, 0:8] = out_bf16 [M, D] x g_b[8, D]^T (bf16, fp32 accumulation): the rank-8 "g = dy (scale B)"
// product of the LoRA adapter on the projection that consumes out_bf16 (peft lora.Linear backward, blip2_mr.py:182-200), for which the
// engine otherwise launches mrblip_lora_rows over the operand this kernel has just written.  nparts = 1: plain dy.
extern "C" int mrblip_rmsnorm_bwd_parts_g(const float* dy, long long lddy, int nparts, long long pstride, int ext_part, uint32_t ext_site, float ext_p,
                                          const float* x, long long ldx, const float* weight, int M, int D, float eps, const float* dx_add, long long ldadd,
                                          float* dx, long long lddx, void* out_bf16, long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                                          const void* g_b, long long ldgb, void* g_out, long long ldg, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  MRB_REQUIRE(nparts >= 1 && nparts <= 32 && (pstride % 4) == 0 && (lddy % 4) == 0 && ((uintptr_t)dy % 16) == 0, "rmsnorm_bwd_parts_g: 1..32 parts of 16-B aligned rows");
  MRB_REQUIRE(out_bf16 && (ldob % 4) == 0, "rmsnorm_bwd_parts_g: bf16 output missing / unaligned");
  MRB_REQUIRE(g_b && g_out && (ldgb % 4) == 0 && ((uintptr_t)g_b % 8) == 0 && (ldg % 8) == 0 && ((uintptr_t)g_out % 16) == 0 && ldgb >= D,
              "rmsnorm_bwd_parts_g: g_b is [8, >= D] with 8-B aligned rows, g_out [M, >= 8] with 16-B aligned rows");
  MRB_REQUIRE(!(p_drop > 0.f || ext_p > 0.f) || seed_ptr, "rmsnorm_bwd_parts_g: dropout needs a device seed pointer");
  DropoutArg d, de;
  d.seed_ptr = (p_drop > 0.f) ? seed_ptr : nullptr; d.site = site; d.thresh24 = (uint32_t)(p_drop * 65536.0f + 0.5f); d.inv_keep = 1.0f / (1.0f - p_drop);
  de.seed_ptr = (ext_p > 0.f) ? seed_ptr : nullptr; de.site = ext_site; de.thresh24 = (uint32_t)(ext_p * 65536.0f + 0.5f); de.inv_keep = 1.0f / (1.0f - ext_p);
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL((norm_bwd_kernel<true, true>), dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, weight, M, D, eps, dx_add, ldadd, dx, lddx, (float*)nullptr,
                     (float*)nullptr, (bf16_t*)out_bf16, ldob, d, (float*)nullptr, nparts, pstride, ext_part, de, (const bf16_t*)g_b, ldgb, (bf16_t*)g_out, ldg);
  return mrblip_check_launch("rmsnorm_bwd_parts_g");
}

extern "C" int mrblip_rmsnorm_bwd(const float* dy, long long lddy, const float* x, long long ldx, const float* weight, int M, int D,
                                  float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, hipStream_t stream) {
  if (int e = norm_check(M, D, x, ldx)) return e;
  const int grid = min((M + 3) / 4, 2048);
  hipLaunchKernelGGL(norm_bwd_kernel<true>, dim3(grid), dim3(256), 0, stream, dy, lddy, x, ldx, weight, M, D, eps, dx_add, ldadd, dx, lddx, (float*)nullptr, (float*)nullptr);
  return mrblip_check_launch("rmsnorm_bwd");
}
