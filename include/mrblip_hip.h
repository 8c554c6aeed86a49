/* libmrblip_hip.so — C ABI of the MI355X-native (gfx950) Mr. BLIP / Chrono train-step kernels.
 *
 * The reference (sudo-Boris/mr-Blip) is pure Python/PyTorch and has no FFI of its own; each entry point below
 * replaces the torch call sites cited next to it (file:line relative to the reference tree) and is what a
 * ctypes / pybind stub on the reference side would bind.
 *
 * Conventions: every function returns 0 on success or a negative code (message: mrblip_last_error(), thread
 * local).  All buffers are device memory owned by the caller; the library never allocates, never synchronises,
 * never changes the current device, and launches only on `stream`.  bf16 tensors are raw uint16 bits.
 * Leading dimensions / strides are in ELEMENTS.  Dropout: mask = f(index, *seed_ptr, site) with a counter hash
 * (csrc/common.h mrb_hash), p == 0 disables it; *seed_ptr is read on the device so captured hipGraphs replay
 * with fresh masks after mrblip_seed_bump.
 */
#ifndef MRBLIP_HIP_H
#define MRBLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mrblip_stream_t; /* hipStream_t */

const char* mrblip_last_error(void);
int mrblip_abi_version(void);

/* C[M,N] = A[M,K] W[N,K]^T (+ Aext[M,64] Wext[N,64]^T : LoRA K-extension) with fused epilogue
 *   v = acc + bias;  out2 = bf16(v) (optional pre-activation);  v = gelu_erf(v) if act==1;  v = dropout(v);
 *   out = residual + v   (fp32 or bf16 out).
 * act == 2 (round 6, GELU BACKWARD: Qformer.py:349-360 backward): out2 is READ — the forward's saved pre-activation — and
 *   out = bf16(bf16(acc) * gelu_erf'(out2)), the same bits as a bf16 GEMM followed by mrblip_gelu_bwd; bf16 out, no bias / residual /
 *   dropout / K extension, generic tile forms only.
 * gated != 0: W = [wi_0; wi_1] stacked ([N = 2*Nh, K]); out[M,Nh] = dropout(gelu(h0) * h1), out2 = [h0 | h1].
 * tile_cfg bits 0..7 = tile form: 0 auto; 1 = 256x256 / 8 waves; 2 = 128x128 / 4 waves; 3 = skinny-M weight-streaming kernel; 4 = 64x128;
 *   5 = 64x64; 6 = 256x256 / 4 waves (compiler-scheduled); 7 = 256x128 BK=32 3-stage; 8 = 256x256 / 16 waves (persistent); 9 = 128x128 / 8 waves;
 *   10 = 256x128 / 16 waves; 11 = 128x256 / 16 waves; 12 = 256x192 / 8 waves; 13 = 256x256 / 4 waves of 128x128, hand-pipelined K loop
 *   (plain epilogues: the frozen-ViT GEMMs); 14 = the same at 256x192; 15 = 13 with the epilogue deferred into the next tile's MFMA
 *   shadow (bf16 out, bias, optional GELU); 16 = 13's tile on 16x16x32 MFMAs (bit-identical results).  Auto picks among 2-5, 8, 13;
 *   the others are measured variants kept selectable.
 * tile_cfg bits 8..16 = CU reserve of the persistent forms 13 - 16: CUs (a multiple of 8, one per XCD) this launch leaves to other
 *   streams (per-call; 0 = the calling thread's default, see mrblip_gemm_set_cu_reserve).
 * tile_cfg bits 17..20 = K split of the skinny form (M <= 32 rows, fp32 out, no residual / out2 / act): that many blocks share an
 *   output tile along K and atomically ADD their partial products (bias once, the output dropout mask on every partial) to `out`,
 *   which the caller pre-initialised (mrblip_lora_rows_init) — puts every CU on the weight stream of a 12-token decoder GEMM.
 * Replaces F.linear / nn.Linear / Conv2d(k=s=14): eva_vit.py:120-126,146,54-61,196-203; Qformer.py:141-147,
 * 285-289,349-375; modeling_t5.py:323-329,536-560,1870; blip2_mr.py:491 (t5_proj); peft LoRA Linear. */
int mrblip_gemm_bf16(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                     const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                     void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act, int gated,
                     const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg, mrblip_stream_t stream);

/* The same entry on IEEE fp16 operands (A, W; a 16-bit `out` is fp16 too): v_mfma_f32_32x32x16_f16, fp32 accumulate.  The reference's GPU
 * arithmetic for the frozen ViT is fp16 autocast over fp16 weights (blip2_mr.py:446; eva_vit.py:397-412, 439-441): 3 more mantissa bits
 * than bf16 at the same MFMA rate.  Only the 4-wave 256x256 kernel (tile_cfg 0 / 13) has this form: plain epilogues — bias, bias + GELU
 * (16-bit out), fp32 out with or without the fp32 residual.  Replaces F.linear under fp16 autocast: eva_vit.py:120-126, 146, 54-61, 196-203. */
int mrblip_gemm_f16(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext,
                    const void* Wext, long long ldwext, int M, int N, int K, void* out, long long ldo, int out_f32,
                    void* out2, long long ldo2, const float* bias, const float* residual, long long ldr, int act, int gated,
                    const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg, mrblip_stream_t stream);

/* LayerNorm / T5 RMSNorm over rows of fp32 x[M,D] (D <= 2048, D % 4 == 0); fp32 statistics.
 * eva_vit.py:157,163; blip2.py:113-119 (ln_vision); Qformer.py:104-107,285-289,372-375; modeling_t5.py:254-277. */
int mrblip_layernorm_fwd(const float* x, long long ldx, const float* gamma, const float* beta, int M, int D, float eps,
                         void* out_bf16, long long ldob, float* out_f32, long long ldof, mrblip_stream_t stream);
/* LayerNorm with an fp16 16-bit output: the fp16-operand ViT's norm1 / norm2 (eva_vit.py:157-163: fp32 LayerNorm, fp16 consumer) */
int mrblip_layernorm_fwd_f16(const float* x, long long ldx, const float* gamma, const float* beta, int M, int D, float eps,
                             void* out_f16, long long ldob, float* out_f32, long long ldof, mrblip_stream_t stream);
int mrblip_rmsnorm_fwd(const float* x, long long ldx, const float* weight, int M, int D, float eps, void* out_bf16,
                       long long ldob, float* out_f32, long long ldof, mrblip_stream_t stream);
/* Workspace of the ORDERED reductions (round 6).  mrblip_cross_entropy (row terms), mrblip_colsum (row-block partial sums) and
 * mrblip_layernorm_bwd with dgamma (block partial sums) add their partials in a FIXED order — the block that draws the last ticket does it —
 * so a train step is bit-reproducible.  Scratch and tickets are the CALLER's: register, per calling thread, `bytes` >=
 * mrblip_reduce_workspace_bytes() of ZEROED, 16-B aligned device memory (the kernels return the tickets to zero); the registration stays
 * until replaced (ws = NULL removes it).  Launches that share one workspace must be stream-ordered — give every stream its own
 * (mrblip/ops.py keeps one per (device, stream)).  Without a workspace the three kernels fall back to fp32 atomics (arrival order).
 * The library itself keeps no device-side or process-global mutable state. */
int mrblip_set_reduce_workspace(void* ws, long long bytes);
long long mrblip_reduce_workspace_bytes(void);
/* dx = dx_add + dLN(dy); optional dgamma/dbeta += the row sums (ordered through the reduce workspace, else fp32 atomics) */
int mrblip_layernorm_bwd(const float* dy, long long lddy, const float* x, long long ldx, const float* gamma, int M, int D,
                         float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, float* dgamma,
                         float* dbeta, mrblip_stream_t stream);
/* layernorm_bwd (dx only) that also writes bf16(dropout-backward(dx)) — the next GEMM's operand — in the same launch; bit-identical to
 * mrblip_layernorm_bwd + mrblip_cast_dropout (Q-Former post-LayerNorm sub-layers: Qformer.py:285-289, 372-375) */
int mrblip_layernorm_bwd_cast(const float* dy, long long lddy, const float* x, long long ldx, const float* gamma, int M, int D, float eps,
                              const float* dx_add, long long ldadd, float* dx, long long lddx, void* out_bf16, long long ldob,
                              const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);
int mrblip_rmsnorm_bwd(const float* dy, long long lddy, const float* x, long long ldx, const float* weight, int M, int D,
                       float eps, const float* dx_add, long long ldadd, float* dx, long long lddx, mrblip_stream_t stream);
/* mrblip_rmsnorm_bwd that additionally writes the next GEMM's bf16 operand: out_bf16 = bf16(dropout-backward(dx)) for the mask of
 * (seed, site, p) over element index row * D + col (bit-identical to a following mrblip_cast_dropout launch).  The T5 block backward:
 * T5LayerNorm backward (modeling_t5.py:239-262) followed by the previous sub-layer's nn.Dropout backward (:613-615, :345-347). */
int mrblip_rmsnorm_bwd_cast(const float* dy, long long lddy, const float* x, long long ldx, const float* weight, int M, int D, float eps,
                            const float* dx_add, long long ldadd, float* dx, long long lddx, void* out_bf16, long long ldob,
                            const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);

/* Fused softmax(Q K^T * scale + bias_lut[h][clamp(k-q,-128,128)+128] + masks) V, fp32 online softmax, bf16 I/O.
 * strides = {batch, head, row} in elements (rows are contiguous in head_dim).  Vt/Kt/Qt/dOt are
 * [B,H,roundup32(D),roundup32(S)] transposed zero-padded copies from mrblip_head_transpose.  LSE/Delta are
 * [B,H,roundup32(Sq)] fp32.  eva_vit.py:128-145; Qformer.py:195-262; modeling_t5.py:392-472,536-603.
 * drop_bits (optional, with dropout): uint32 [B*H, roundup32(Sk)/32, roundup32(Sq)] scratch that carries the keep mask of the
 * probability dropout from the forward to the backward (head_dim 64, non-causal, Sq > 32 shapes; ignored elsewhere) so the
 * backward does not recompute the counter hash; pass the SAME buffer (or NULL to both) to fwd and bwd of one attention. */
int mrblip_attention_fwd(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* Vt,
                         void* O, const long long* o_strides, float* LSE, int B, int H, int Sq, int Sk, int D, float scale,
                         const float* bias_lut, const int* kmask, int causal, const uint32_t* seed_ptr, uint32_t site,
                         float p_drop, uint32_t* drop_bits, mrblip_stream_t stream);
/* The plain (frozen-ViT) forward with V read ROW-major, straight from the fused qkv projection output — no transposed copy of V
 * (the V^T fragments are gathered from the row-major LDS image with ds_read_b64_tr_b16).  head_dim in (64, 96], Sq > 32, no bias /
 * mask / dropout.  eva_vit.py:128-145 (Attention.forward: q, k, v = qkv[0], qkv[1], qkv[2]; softmax(q k^T * scale) v). */
int mrblip_attention_fwd_rowv(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* V,
                              const long long* v_strides, void* O, const long long* o_strides, float* LSE, int B, int H, int Sq, int Sk,
                              int D, float scale, mrblip_stream_t stream);
/* ... on IEEE fp16 Q / K / V / O (fp16 matmuls, fp32 softmax: eva_vit.py:128-145 under the reference's fp16 autocast, blip2_mr.py:446) */
int mrblip_attention_fwd_rowv_f16(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* V,
                                  const long long* v_strides, void* O, const long long* o_strides, float* LSE, int B, int H, int Sq, int Sk,
                                  int D, float scale, mrblip_stream_t stream);
int mrblip_attention_bwd(const void* Q, const long long* q_strides, const void* K, const long long* k_strides, const void* V,
                         const long long* v_strides, const void* O, const long long* o_strides, const void* dO,
                         const long long* do_strides, const void* Kt, const void* Qt, const void* dOt, const float* LSE,
                         float* Delta, void* dQ, const long long* dq_strides, void* dK, const long long* dk_strides, void* dV,
                         const long long* dv_strides, int B, int H, int Sq, int Sk, int D, float scale, const float* bias_lut,
                         const int* kmask, int causal, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                         const uint32_t* drop_bits, mrblip_stream_t stream);
/* Spad: padded row length of the transposed copy (multiple of 32), 0 = roundup32(S) */
int mrblip_head_transpose(const void* src, const long long* strides, void* dst, int B, int H, int S, int D, int Spad,
                          const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);

/* Per-THREAD default (thread_local; the library has no process-global mutable state) of the CU reserve that mrblip_gemm_bf16 otherwise
 * takes per call in tile_cfg bits 8..16; returns the calling thread's previous default.  No reference counterpart: the reference runs the ViT forward in line with the
 * rest of forward_mr (blip2_mr.py:287-289); here it runs one clip ahead on a second stream and must not starve the clip being trained. */
int mrblip_gemm_set_cu_reserve(int n_cus);

/* frames fp32 [F,3,IMG,IMG] -> bf16 patch rows [F*(IMG/P)^2, Kpad] in Conv2d weight order (eva_vit.py:196-203) */
int mrblip_patchify(const float* video, void* out_bf16, int F, int IMG, int P, int Kpad, mrblip_stream_t stream);
/* the same from uint8 frames [F,3,IMG,IMG] with the processor's ToTensor + Normalize(mean3, std3) applied on the fly
 * (blip_processors.py:63-66; mean3 / std3 are HOST arrays of 3 floats); bit-identical to normalising first */
int mrblip_patchify_u8(const uint8_t* video, const float* mean3, const float* std3, void* out_bf16, int F, int IMG, int P, int Kpad,
                       mrblip_stream_t stream);
/* both with fp16 patch rows (the fp16-operand ViT's patch-embedding GEMM, eva_vit.py:196-203 under fp16 autocast) */
int mrblip_patchify_f16(const float* video, void* out_f16, int F, int IMG, int P, int Kpad, mrblip_stream_t stream);
int mrblip_patchify_u8_f16(const uint8_t* video, const float* mean3, const float* std3, void* out_f16, int F, int IMG, int P, int Kpad,
                           mrblip_stream_t stream);
/* x = [cls ; patches] + pos  (eva_vit.py:328-331) */
int mrblip_vit_assemble(const float* patch, const float* cls, const float* pos, float* x, int F, int NP, int D, mrblip_stream_t stream);
/* dst[dst_idx[i],:] (=|+=) src[src_idx[i],:]; src_idx<0 -> zeros.  Embedding gathers and the frame/timestamp
 * interleave of prompt_concatenation (blip2_mr.py:641-665, 691-757) and their backward. */
int mrblip_row_copy(const float* src, long long lds, const int* src_idx, float* dst, long long ldd, const int* dst_idx, int n_rows,
                    int D, int accumulate, mrblip_stream_t stream);
/* 32 -> 1 frame-token mean (blip2_mr.py:493-498) and its backward */
int mrblip_mean_pool(const float* x, float* out, int F, int n, int D, mrblip_stream_t stream);
int mrblip_mean_pool_bwd(const float* dout, float* dx, int F, int n, int D, mrblip_stream_t stream);
int mrblip_cast_dropout(const float* x, long long ldx, void* out_bf16, long long ldob, float* out_f32, long long ldof, int M, int N,
                        const uint32_t* seed_ptr, uint32_t site, float p, mrblip_stream_t stream);
int mrblip_gelu_bwd(const void* dy, const void* h, void* dh, long long n, mrblip_stream_t stream);
int mrblip_gated_gelu_bwd(const void* dy, long long lddy, const void* h, long long ldh, void* dh, long long lddh, int M, int Nh,
                          const uint32_t* seed_ptr, uint32_t site, float p, mrblip_stream_t stream);
/* CrossEntropyLoss(ignore_index=-100, mean) + dlogits (modeling_t5.py:1873-1877); *loss += the row terms in ROW order through the reduce
 * workspace (mrblip_set_reduce_workspace; R <= 4096), else by fp32 atomics */
int mrblip_cross_entropy(const float* logits, long long ldl, const int* labels, int R, int V, float inv_count, float* loss,
                         void* dlogits_bf16, long long ldd, mrblip_stream_t stream);
/* the same with the count of valid rows in DEVICE memory (inv_count = 1 / max(*n_valid, 1), fp32): a captured graph (hipGraph) of the step
 * then serves batches with any number of valid label tokens */
int mrblip_cross_entropy_nvalid(const float* logits, long long ldl, const int* labels, int R, int V, const int* n_valid, float* loss,
                                void* dlogits_bf16, long long ldd, mrblip_stream_t stream);
/* torch.optim.AdamW step on a flat segment; hyper = {lr, 1/bias_corr1, 1/sqrt(bias_corr2), grad_scale} on device
 * (runner_base.py:102-132, moment_retrieval.py:221-233) */
int mrblip_adamw(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2, float eps,
                 float weight_decay, mrblip_stream_t stream);
int mrblip_seed_bump(uint32_t* seed, mrblip_stream_t stream);
/* Reads [ptr, ptr + bytes) with n_blocks small workgroups and drops the data: pulls the next launch's weights into the memory-side cache
 * from a side stream.  No torch counterpart (the reference leaves weight residency to the caches); it changes no result. */
/* The same from inside a GEMM launch: the calling thread's NEXT mrblip_gemm_bf16 / mrblip_gemm_lora_dx launch starts n_blocks extra
 * workgroups that read the range (and a second one, ptr2 / bytes2, may be NULL / 0) while the tiles compute (no side stream, no event).  One-shot; ignored by the tile forms that have
 * no such role. */
int mrblip_gemm_set_prefetch(const void* ptr, long long bytes, const void* ptr2, long long bytes2, int n_blocks);
/* One-shot: the calling thread's NEXT mrblip_gemm_bf16 launch (generic tile kernel, M > 64, K extension read last) computes its own
 * K-extension operand Aext[m, 0:R] = dropout(A)[m, 0:K] acat^T — what mrblip_lora_rows would have written in a launch of its own (peft
 * lora_A(lora_dropout(x)), same bits) — in its first workgroups while the tiles already run; the tiles wait for flags[m / 16] == epoch before
 * they read those rows.  flags: >= ceil(M / 16) + 4 ZERO-initialised words shared by the launches of ONE stream: one flag per 16 rows, then
 * (words n_flags - 3, n_flags - 2) the ticket and finished-workgroup counters by which the launch hands out its roles in the order its
 * workgroups START (round 5: producers are then running before any consumer can wait for them, whatever the dispatch order; the kernel
 * returns both to zero), then the fallback error word (non-zero after a tile's bounded wait ran out — it never happens in a correct
 * run; `err` below overrides where it lives); epoch: a value no earlier launch left in the flags.
 * The mask uses the GEMM call's seed pointer with call-site id `site`.  The epoch is a launch argument: a captured graph that replays such a
 * launch must clear the flag words between replays (one memset node), or every replay would find the previous replay's flags set. */
int mrblip_gemm_set_thin(const void* acat, long long lda, int R, int K, uint32_t site, float p_drop, uint32_t* flags, long long n_flags,
                         uint32_t epoch, uint32_t* err /* NULL: the last flag word */);
/* Failing loudly (round 5).  `err` above is the word a timed-out tile sets; the host side keeps ONE per device, hands it to every launch
 * and to mrblip_adamw_guarded: a non-zero *guard makes the optimizer step a no-op on the device (the reference has no counterpart: its
 * GEMMs cannot time out; the closest is GradScaler's skipped step on inf / nan, runner_base.py:127-132, moment_retrieval.py:221-233),
 * and the engine reads the word one step late and raises.  mrblip_gemm_debug_stall_thin(1) is a TEST HOOK: the calling thread's later
 * launches start thin-role workgroups that exit without publishing, which drives every consumer into its bounded wait; returns the
 * previous setting. */
int mrblip_adamw_guarded(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2, float eps,
                         float weight_decay, const uint32_t* guard, mrblip_stream_t stream);
int mrblip_gemm_debug_stall_thin(int on);
int mrblip_prefetch(const void* ptr, long long bytes, int n_blocks, mrblip_stream_t stream);
/* LoRA r=8 (peft 0.13.0 Linear; blip2_mr.py:182-200,236).  The rank-8 products themselves run on mrblip_gemm_bf16
 * (u = drop(x) Acat^T, g = dy Bblk^T, dBt += u^T dy, dA += g^T drop(x) on transposed copies); these are the side pieces:
 *   dx[m,k] += drop_mask(m,k) * sum_{r<R} G[m,r] Acat[r,k]   ;   out = dropout(x) bf16 -> bf16 (lora_dropout) */
int mrblip_lora_dx_add(void* dx, long long lddx, int dx_f32, const void* G, long long ldg, const void* Acat_bf16, int R, int M, int K,
                       const uint32_t* seed_ptr, uint32_t site, float p, mrblip_stream_t stream);
int mrblip_dropout_bf16(const void* x, long long ldx, void* out, long long ldo, int M, int N, const uint32_t* seed_ptr, uint32_t site,
                        float p, mrblip_stream_t stream);
/* out[c] += sum_m x[m,c] (bias gradient of t5_proj, blip2_mr.py:270-272); row blocks added in block order through the reduce workspace
 * (N <= 8192), else by fp32 atomics */
int mrblip_colsum(const float* x, long long ldx, int M, int N, float* out, mrblip_stream_t stream);
/* LoRA weight gradients without transposed copies: for j < R/8: outs[j][(r%8)*lds[j] + c - col0[j]] += sum_m U[m, 8j + r%8] * drop(Y)[m, c]
 * for the columns col0[j] <= c < col0[j] + ncols[j]  (dB^T = u^T dy is block-diagonal over the adapters of a fused group, dA = g^T dropout(x)) */
int mrblip_lora_tn(const void* Y, long long ldy, const void* U, long long ldu, int M, int C, int R, float* const* outs, const int* col0,
                   const int* ncols, const long long* lds, const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);
/* both weight gradients of a fused LoRA group in one launch: dBt_j += U_j^T dY (adapter j's output columns), dA_j += G_j^T dropout(X) */
int mrblip_lora_grads(const void* dY, long long lddy, const void* U, long long ldu, const void* X, long long ldx, const void* G,
                      long long ldg, int M, int N, int K, int R, float* const* dBt, const int* b_col0, const int* b_ncols,
                      const long long* b_lds, float* const* dA, const long long* a_lds, const uint32_t* seed_ptr, uint32_t site,
                      float p_drop, mrblip_stream_t stream);
/* The same for SEVERAL fused groups in ONE launch (round 5: every group of a T5 layer — peft computes these one Linear at a time in
 * autograd, blip2_mr.py:182-200): jobs[i] holds the arguments of one mrblip_lora_grads call (arrays of 4: unused slots NULL / 0), all
 * with the same device seed; 1..8 jobs; same block -> output ownership as the one-group launch, i.e. the same bits. */
typedef struct MrblipLoraGradsJob {
  const void* dY; long long lddy; const void* U; long long ldu; const void* X; long long ldx; const void* G; long long ldg;
  int M, N, K, R;
  float* dBt[4]; int b_col0[4]; int b_ncols[4]; long long b_lds[4];
  float* dA[4]; long long a_lds[4];
  uint32_t site; float p_drop;
} MrblipLoraGradsJob;
int mrblip_lora_grads_batched(int n_jobs, const MrblipLoraGradsJob* jobs, const uint32_t* seed_ptr, mrblip_stream_t stream);
/* LoRA "down" product with the lora_dropout fused into the operand load: U[M,N] = dropout(X)[M,K] Acat[N,K]^T (bf16; peft lora_A(lora_dropout(x))) */
int mrblip_gemm_lora_down(const void* X, long long ldx, const void* Acat, long long lda, int M, int N, int K, void* U, long long ldu,
                          const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);
/* The same product (and the backward's g = dY B: p_drop = 0) as a ROW kernel: the R <= 32 thin vectors A[R,K] live in LDS, one wavefront
 * reads a row of X once in full 16-B pieces, masks it in registers and takes R dot products (v_dot2c_f32_bf16); U[M, 0:R] bf16.
 * Replaces peft's lora_A(lora_dropout(x)) / grad_output @ lora_B.weight of every adapted Linear (blip2_mr.py:182-200).
 * seg: NULL, or 2 * R/8 ints [k0, k1) per 8-row group of A outside of which those rows are zero (block-diagonal B^T of a fused group). */
int mrblip_lora_rows(const void* X, long long ldx, const void* A, long long lda, int M, int R, int K, void* U, long long ldu,
                     const int* seg, const uint32_t* seed_ptr, uint32_t site, float p_drop, mrblip_stream_t stream);
/* mrblip_lora_rows plus a side job over the same M rows: init_dst[m, 0:init_n] = init_src ? init_src[m, :] : 0 (fp32) — the pre-initialised
 * output (residual or zero) that a following K-split mrblip_gemm_bf16 adds its partial products to; no launch of its own. */
int mrblip_lora_rows_init(const void* X, long long ldx, const void* A, long long lda, int M, int R, int K, void* U, long long ldu,
                          const int* seg, const uint32_t* seed_ptr, uint32_t site, float p_drop, float* init_dst, long long ld_idst,
                          const float* init_src, long long ld_isrc, int init_n, mrblip_stream_t stream);
/* mrblip_lora_dx_add for `groups` LoRA groups that share one input, in ONE pass over dx (round 4): dx[m, k] += sum_g mask_g[m, k] *
 * (G[m, g * g_gstride : + R] A_g[R, K])[k], A_g at A + g * a_gstride elements, mask_g = the forward's lora_dropout keep mask of group g
 * (call-site id site0 + g * site_stride) scaled by 1 / (1 - p): the rank-R input-gradient terms of all decoder layers' cross K / V adapters. */
int mrblip_lora_dx_add_batched(float* dx, long long lddx, const void* G, long long ldg, long long g_gstride, const void* A, long long a_gstride, int R,
                               int M, int K, int groups, const uint32_t* seed_ptr, uint32_t site0, uint32_t site_stride, float p, mrblip_stream_t stream);
/* mrblip_lora_rows for `groups` problems in ONE launch (round 4): group g reads X + g * x_gstride (0: the same rows for every group), A + g *
 * a_gstride, writes U + g * u_gstride (strides in elements) and draws its mask with call-site id site0 + g * site_stride — the LoRA "down"
 * products of all decoder layers' EncDecAttention.k / .v adapters on one encoder output, and their backward's g = dy B on the layers' dy
 * column blocks (peft lora.Linear around modeling_t5.py:561-599).  K %% 32 == 0, R <= 32. */
int mrblip_lora_rows_batched(const void* X, long long ldx, long long x_gstride, const void* A, long long lda, long long a_gstride, int M, int R, int K,
                             void* U, long long ldu, long long u_gstride, int groups, const uint32_t* seed_ptr, uint32_t site0, uint32_t site_stride,
                             float p_drop, mrblip_stream_t stream);
/* T5LayerNorm (modeling_t5.py:254-277) fused with the LoRA "down" product of its output: out_bf16 = bf16(x * rsqrt(mean(x^2) + eps) * weight),
 * U[M, 0:R] = dropout(out_bf16) A[R,D]^T — one launch for the norm-fed adapted projections (q/k/v, wi_0/wi_1, EncDecAttention.q). */
int mrblip_rmsnorm_lora_fwd(const float* x, long long ldx, const float* weight, int M, int D, float eps, void* out_bf16, long long ldob,
                            const void* A, long long lda, int R, void* U, long long ldu, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                            mrblip_stream_t stream);
/* One launch for an adapted projection of few rows — the T5 DECODER's, or a short encoder's (R <= 16; <= 80 without x32 / mode 2; replaces mrblip_lora_rows / mrblip_rmsnorm_lora_fwd + mrblip_gemm_bf16
 * of peft's lora.Linear.forward around modeling_t5.py:449-603 (q/k/v/o), :323-329 (wi_0/wi_1/wo), and mrblip_lora_rows + mrblip_gemm_lora_dx of
 * its backward):  xin = x32 ? bf16(RMSNorm(x32) * gamma) (saved to xin) : xin;  U[R, 0:Rk] = bf16(dropout(xin; in_site, in_p) A[Rk,K]^T);
 * acc = xin W[N,K]^T + (ext_p > 0 ? mask(ext_site, ext_p) (.) : ) U Bt[N,64]^T;
 * mode 0: out bf16 = acc;  mode 1: out fp32 = residual + dropout(acc; out_site, out_p);  mode 2 (W, Bt hold 2 N rows: wi_0 then wi_1):
 * out2 bf16 [R, 2 N] = [acc_0 | acc_1], out bf16 = dropout(gelu(acc_0) * acc_1).  N % 16 == 0, K % 32 == 0, Rk in {8, 16, 24, 32}. */
int mrblip_dec_proj(const float* x32, long long ldx32, const float* gamma, float eps, void* xin, long long ldxin, const void* W, long long ldw,
                    const void* A, long long lda, int Rk, const void* Bt, long long ldbt, void* U, long long ldu, int R, int N, int K, int mode,
                    void* out, long long ldo, const float* residual, long long ldr, void* out2, long long ldo2, const uint32_t* seed_ptr,
                    uint32_t in_site, float in_p, uint32_t out_site, float out_p, uint32_t ext_site, float ext_p,
                    /* mode 0, optional: head-transposed copies (mrblip_head_transpose layout [B, H, 64, t_spad]) of up to three consecutive column
                     * ranges of width t_inner; rows are b * t_rows + s */
                    void* tout0, void* tout1, void* tout2, int t_inner, int t_rows, int t_spad, long long t_bs, long long t_hs,
                    mrblip_stream_t stream);
/* Cross-block key split of the few-query attention form (Sq <= 32, no bias LUT, head_dim 64: the T5 decoder's cross attention,
 * modeling_t5.py:536-603 with 8-14 label rows against ~2000 encoder positions; round 4; host-side state of the calling thread, no torch
 * counterpart).  With a workspace registered, mrblip_attention_fwd and the dQ pass of mrblip_attention_bwd cut the key range of such a
 * problem into n_split chunks of whole 32-key tiles, one block each (0 = about one block per CU); partial (m, l, O) / partial dQ meet in
 * the workspace (write-through stores + a ticket per (batch, head)) and the last arriver combines them in chunk order — deterministic,
 * independent of dispatch order.  ws: >= 16 KB + B * H * n_split * 9216 bytes of 16-B aligned device memory whose first 16 KB are ZERO
 * (the tickets; the kernels leave them zero); launches that use it must be stream-ordered.  ws = NULL: the one-block-per-head form. */
int mrblip_attention_set_split_workspace(void* ws, long long bytes, int n_split);
/* Drops every pending one-shot of the calling thread (mrblip_gemm_set_extra / _set_prefetch / _set_thin) without launching anything: the
 * exception path of a caller that fails between a setter and its GEMM (every dispatch also consumes them, whether it launches or not). */
int mrblip_gemm_clear_one_shots(void);
/* One-shot extras of the calling thread's NEXT mrblip_gemm_bf16 / mrblip_gemm_lora_dx launch (round 4; generic tile kernels only, the call
 * fails loudly for a kernel form that cannot honour them):
 *  - tout0..2 (bf16 output, plain or bias epilogue, heads of 64): head-transposed copies of up to three consecutive column ranges of width
 *    t_inner — exactly what mrblip_head_transpose would write from the output ([B, H, 64, t_spad], rows b * t_rows + s, pad columns zeroed;
 *    one clip, or t_rows %% 32 == 0): q / k / v of a fused projection reach the attention kernels without a transpose launch
 *    (modeling_t5.py:536-560, Qformer.py:141-147 feeding :195-262);
 *  - ext_group_n > 0: output columns [g * ext_group_n, (g + 1) * ext_group_n) take columns [64 g, 64 g + 64) of Aext as their K extension:
 *    the cross-attention K / V projections of ALL decoder layers (modeling_t5.py:561-599, peft LoRA on each) as one GEMM. */
 *    (t_count > 0: range j < t_count goes to tout0 + j * t_stride elements instead of the three pointers);
int mrblip_gemm_set_extra(void* tout0, void* tout1, void* tout2, int t_inner, int t_rows, int t_spad, long long t_bs, long long t_hs,
                          long long t_stride, int t_count, int ext_group_n);
/* Launch shape of mrblip_dec_proj for R <= 16 (round 4; host-side state, no torch counterpart): n_blocks > 0 = blocks of the streaming kernel for
 * the calling thread's later launches — each block owns a contiguous range of 16-column tiles and streams their weight rows back to back
 * (0 = one block per CU, < 0 = unchanged); the engine asks for as many blocks as the frozen-ViT look-ahead leaves CUs.  version 0 = the
 * one-tile-per-block kernel of round 3, 1 = the streaming kernel (default), any other value = unchanged.  Same results either way
 * (bit-identical: same K split, same summation order).  Returns the previous n_blocks. */
int mrblip_dec_proj_config(int n_blocks, int version);
/* LoRA backward input gradient in one launch: dX[M,N] = dY[M,K] Wt[N,K]^T (+ residual) + mask(site,p) * (G[M,64] AcatT[N,64]^T);
 * N = in_features, K = out_features padded to 64, mask = the forward's lora_dropout keep mask scaled by 1/(1-p) */
int mrblip_gemm_lora_dx(const void* dY, long long lddy, const void* Wt, long long ldwt, const void* G, long long ldg, const void* AcatT,
                        long long ldat, int M, int N, int K, void* dX, long long lddx, int out_f32, const float* residual, long long ldr,
                        const uint32_t* seed_ptr, uint32_t site, float p_drop, int tile_cfg, mrblip_stream_t stream);
/* Round 5: the same input gradient for outputs with too few 256x256 tiles to fill 256 CUs (T5 encoder at ~2000 rows: 64 tiles), as PARTIAL
 * products of a K-split on the hand-pipelined 4-wave kernel: out + s * part_stride = A[:, K range s] W[:, K range s]^T for s < k_splits and,
 * with a K extension (Aext = G [M,64], Wext = AcatT [N,64]), out + k_splits * part_stride = Aext Wext^T — the LoRA term, still UNMASKED.
 * fp32 or bf16 parts, no bias / residual / activation; tile_cfg 13 (256x256 tiles) or 14 (256x192); K %% (64 k_splits) == 0.  The consumer adds
 * the parts in part order and applies the lora_dropout keep mask to the last one:
 *   mrblip_rmsnorm_bwd_parts   = mrblip_rmsnorm_bwd / _bwd_cast with dy = sum of nparts fp32 parts (ext_part: last part masked by (ext_site, ext_p) over [M, D])
 *   mrblip_gated_gelu_bwd_parts = mrblip_gated_gelu_bwd with dy + mask(ext_site, ext_p) (.) dy_ext (two bf16 parts, same leading dimension)
 * Reference: the autograd of peft lora.Linear.forward (blip2_mr.py:182-200) feeding T5LayerNorm / T5DenseGatedActDense backward
 * (modeling_t5.py:254-277, 323-329); same sum in a fixed order, so the step stays bit-reproducible. */
int mrblip_gemm_ksplit(const void* A, long long lda, const void* W, long long ldw, const void* Aext, long long ldaext, const void* Wext,
                       long long ldwext, int M, int N, int K, void* out, long long ldo, long long part_stride, int out_f32, int k_splits,
                       int tile_cfg, mrblip_stream_t stream);
int mrblip_rmsnorm_bwd_parts(const float* dy, long long lddy, int nparts, long long pstride, int ext_part, uint32_t ext_site, float ext_p,
                             const float* x, long long ldx, const float* weight, int M, int D, float eps, const float* dx_add, long long ldadd,
                             float* dx, long long lddx, void* out_bf16, long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                             mrblip_stream_t stream);
/* Round 6: mrblip_rmsnorm_bwd_parts that ALSO computes the rank-8 LoRA product of the operand it writes,
 *   g_out[m, 0:8] = sum_c out_bf16[m, c] * g_b[r, c]      (g_b = scale * B^T of the adapter on the projection that consumes out_bf16, bf16 [8, >= D])
 * i.e. what mrblip_lora_rows(out_bf16, g_b, g_out) would compute in a launch of its own (peft lora.Linear backward: grad of lora_A's output,
 * blip2_mr.py:182-200) — the rows are in this kernel's registers, g_b in LDS, the products on v_dot2c_f32_bf16.  nparts = 1: plain dy. */
int mrblip_rmsnorm_bwd_parts_g(const float* dy, long long lddy, int nparts, long long pstride, int ext_part, uint32_t ext_site, float ext_p,
                               const float* x, long long ldx, const float* weight, int M, int D, float eps, const float* dx_add, long long ldadd,
                               float* dx, long long lddx, void* out_bf16, long long ldob, const uint32_t* seed_ptr, uint32_t site, float p_drop,
                               const void* g_b, long long ldgb, void* g_out, long long ldg, mrblip_stream_t stream);
/* out = (residual, may be NULL or out itself) + part 0 + ... + part nparts-1, fp32, added in part order (the reduce of mrblip_gemm_ksplit for
 * consumers without a parts form: the encoder-output gradient of the stacked cross-attention K / V projections, modeling_t5.py:561-599) */
int mrblip_sum_parts(const float* parts, long long ldp, long long pstride, int nparts, const float* residual, long long ldr, float* out,
                     long long ldo, int M, int N, mrblip_stream_t stream);
int mrblip_gated_gelu_bwd_parts(const void* dy, const void* dy_ext, long long lddy, const void* h, long long ldh, void* dh, long long lddh, int M, int Nh,
                                const uint32_t* seed_ptr, uint32_t site, float p, uint32_t ext_site, float ext_p, mrblip_stream_t stream);
/* fp32 LoRA master weights -> bf16 GEMM operands for every adapter of a device descriptor table (10 int64 per adapter:
 * a_off, bt_off, K, out, acat_off, wext_off, bblk_off, Ntot, acatt_off, 0); acatt = [K,64] transposed copy of scale*A */
int mrblip_lora_pack(const float* flat, void* acat_bf16, void* wext_bf16, void* bblk_bf16, void* acatt_bf16, const long long* desc,
                     int n_adapters, float scale, mrblip_stream_t stream);

/* Round 6: one Q-Former layer's QUERY BRANCH in ONE launch — a workgroup owns one frame's 32 query tokens for the whole layer (every
 * query-side operation is row-wise or confined to one frame, so frames never exchange data): self-attention (qkv dense, 12 heads of 32 x 32,
 * output dense + dropout + residual + LayerNorm), optionally cross-attention over the frame's Tv image tokens (query dense, attention over
 * the caller's precomputed K | V projections `kv` [F * Tv, 1536] and their head-transposed V^T `vt` [F, 12, 64, Tvp], output dense + ...
 * LayerNorm) and the FFN 768 -> 3072 -> 768 (erf-GELU, dropout, residual, LayerNorm).  BERT-base geometry only: 768 features, 12 heads of
 * 64, 32 queries, 3072 intermediate.  Weights are bf16 [N, K] row-major with ld = K (2304x768, 768x768, 768x768, 768x768, 3072x768,
 * 768x3072), vectors fp32.  x_in / x_out: fp32 [F * 32, 768] (the post-LayerNorm hidden state); xb_out: optional bf16 copy of x_out.
 * Saved for the backward, exactly what the launch chain it replaces leaves behind: qkv (bf16 [F * 32, 2304]), o / oc (bf16 attention
 * outputs, ld = ldo), lse / lsec (fp32 [F, 12, 32]), qc (bf16 [F * 32, 768]), y / y2 / y3 (fp32 pre-LayerNorm sums), hpre (bf16 pre-GELU).
 * Dropout (p_drop, *seed_ptr): the draws of the launches it replaces — element dropout on index row * 768 + n per call site, attention
 * draws v3 on (frame * 12 + head) * 32 + query — so a training step is the same function of the seed with either path.
 * Replaces Qformer.py:111-289 (BertSelfAttention + BertSelfOutput), 349-375 (intermediate_query / output_query), 402-474 (BertLayer). */
typedef struct mrblip_qformer_layer {
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
} mrblip_qformer_layer;
int mrblip_qformer_layer_fwd(const mrblip_qformer_layer* layer, mrblip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
