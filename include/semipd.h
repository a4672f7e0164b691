/*
 * semipd.h — C-ABI of libsemipd_hip.so, the MI355X (gfx950) Semi-PD hot path.
 *
 * Every entry point takes plain device pointers, sizes and a hipStream_t
 * (passed as void*, 0 = the legacy default stream).  Nothing here includes a
 * torch type: the shared library is built with hipcc alone.  All functions
 * return 0 on success, a positive hipError_t value when the HIP runtime
 * failed, or a negative SEMIPD_E* code for argument errors.  No entry point
 * synchronises the stream or allocates memory, so every launch is
 * hipGraph-capturable (reference requirement: base_attn_backend.py:22-55).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the reference checkout).
 */
#ifndef SEMIPD_H_
#define SEMIPD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types of activations / KV rows (a subset of at::ScalarType).  The two fp8 types are
 * KV-POOL STORAGE types only (OCP e5m2 / e4m3fn = torch.float8_e5m2 / float8_e4m3fn,
 * `--kv-cache-dtype fp8_e5m2|fp8_e4m3`, mem_cache/memory_pool.py:205-209): rows are rounded to nearest
 * even when stored and expanded exactly to the activation type when read; all arithmetic stays in the
 * activation type.  Entry points that touch the pool take `kv_dtype` (== dtype for plain rows). */
enum { SEMIPD_F32 = 0, SEMIPD_F16 = 1, SEMIPD_BF16 = 2, SEMIPD_F8E5M2 = 3, SEMIPD_F8E4M3 = 4 };

/* argument-error codes (negative; HIP errors are returned positive) */
enum {
  SEMIPD_EINVAL = -1,      /* bad size / null pointer                */
  SEMIPD_EDTYPE = -2,      /* unsupported element type               */
  SEMIPD_ESHAPE = -3,      /* head size / group size not supported   */
  SEMIPD_EALIGN = -4,      /* pointer or stride not 16-byte aligned  */
  SEMIPD_ENOTFOUND = -5,   /* IPC mapping not known                  */
};

/* Library version and last error string (thread-local). */
int semipd_version(void);
const char* semipd_last_error(void);

/* ------------------------------------------------------------------ */
/* a1  RMSNorm                                                         */
/* ------------------------------------------------------------------ */
/* out[t,:] = in[t,:] * rsqrt(mean(in[t,:]^2)+eps) * w   (fp32 math)
 * replaces torch.ops.sgl_kernel.rmsnorm
 *   (sgl-kernel/csrc/torch_extension.cc:49, python/sgl_kernel/elementwise.py:9-19;
 *    semantics layers/layernorm.py:59-76).  in_stride/out_stride in elements. */
int semipd_rmsnorm(void* out, const void* in, const void* weight, int64_t num_tokens,
                   int64_t hidden, int64_t in_stride, int64_t out_stride, float eps, int dtype,
                   void* stream);

/* x = in + res (fp32); res = x; in = x*rsqrt(mean(x^2)+eps)*w; both in place.
 * replaces torch.ops.sgl_kernel.fused_add_rmsnorm
 *   (torch_extension.cc:52; csrc/elementwise/fused_add_rms_norm_kernel.cu:24-55). */
int semipd_fused_add_rmsnorm(void* inout, void* residual, const void* weight, int64_t num_tokens,
                             int64_t hidden, float eps, int dtype, void* stream);

/* ------------------------------------------------------------------ */
/* activation                                                          */
/* ------------------------------------------------------------------ */
/* out[t,:d] = silu(in[t,:d]) * in[t,d:2d]
 * replaces torch.ops.sgl_kernel.silu_and_mul (torch_extension.cc:61;
 *   csrc/elementwise/activation.cu:36-50; layers/activation.py:41-51). */
int semipd_silu_and_mul(void* out, const void* in, int64_t num_tokens, int64_t d, int dtype,
                        void* stream);

/* ------------------------------------------------------------------ */
/* a2/a3  RoPE and KV-pool store                                       */
/* ------------------------------------------------------------------ */
/* In-place rotary embedding of q [T,Hq,head] and k [T,Hk,head] rows
 * (row strides in elements), cos_sin_cache fp32 [max_pos, rot_dim]
 * (cos first half, sin second half), positions int64 [T].
 * interleave=0 -> neox pairing (i, i+rot/2); 1 -> GPT-J pairing (2i, 2i+1).
 * replaces torch.ops.sgl_kernel.apply_rope_pos_ids_cos_sin_cache
 *   (torch_extension.cc:70-73; csrc/elementwise/rope.cu:22-89;
 *    layers/rotary_embedding.py:113-169). */
int semipd_rope_inplace(void* q, void* k, const float* cos_sin_cache, const int64_t* positions,
                        int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size,
                        int rot_dim, int64_t q_stride, int64_t k_stride, int interleave, int dtype,
                        void* stream);

/* Same rotation on head slices that are not densely packed: q / k point at the first rotary element
 * of head 0, token and head strides in elements.  Used by MLA, where q_pe = q[..., 128:] (head stride
 * 192) and k_pe = latent[..., 512:] (one 64-wide "head" in a 576-wide row), GPT-J pairing
 * (replaces DeepseekScalingRotaryEmbedding.forward, layers/rotary_embedding.py:710-748). */
int semipd_rope_inplace_strided(void* q, void* k, const float* cos_sin_cache, const int64_t* positions,
                                int64_t num_tokens, int num_q_heads, int num_k_heads, int rot_dim,
                                int64_t q_token_stride, int64_t q_head_stride, int64_t k_token_stride,
                                int64_t k_head_stride, int interleave, int dtype, void* stream);

/* Fused a2+a3: rotate q in place, rotate k, and scatter rotated k and v into
 * the paged pool rows k_buf[loc[t]], v_buf[loc[t]] (row = Hk*head elements,
 * pool row stride in elements).  k itself is also updated in place so callers
 * that keep k_extend contiguous (extend attention) see rotated keys.
 * replaces RotaryEmbedding.forward_cuda + MHATokenToKVPool.set_kv_buffer
 *   (layers/rotary_embedding.py:143-169; mem_cache/memory_pool.py:316-346). */
int semipd_rope_kv_store(void* q, void* k, const void* v, void* k_buf, void* v_buf,
                         const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                         int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size,
                         int v_head_size, int rot_dim, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, int64_t kbuf_stride, int64_t vbuf_stride, int interleave,
                         int dtype, int kv_dtype, void* stream);

/* MLA decode step (DeepseekV2AttentionMLA.forward_absorb): everything between the merged [q_proj | kv_a_proj_with_mqa] GEMM
 * and the attention kernel in one launch.  planes [n_planes][num_tokens][Hq * (nope + rope) + lora + rope] fp32 are the
 * K-slice planes of that GEMM (semipd_stream_linear_planes on the concatenated weight); each value is summed in slice order
 * and rounded to dtype (the GEMM's own reduction), then: q_nope -> q_nope_out [num_tokens, Hq, nope] (dense); q_pe rotated
 * (GPT-J pairs, cos_sin_cache [max_pos, rope] fp32) -> q_input[t, h, lora : lora + rope] (strides in elements); the latent
 * normalised (RMSNorm * norm_weight, eps) and k_pe rotated -> kv_buf row loc[t] (lora + rope columns, kv_dtype = dtype or
 * an fp8 pool type).  Same bits as the separate reduction, semipd_rmsnorm, semipd_rope_strided, the q_pe copy and
 * semipd_kv_store_cvt.
 * replaces kv_a_layernorm + rotary_emb + q_input[..., lora:] = q_pe + set_kv_buffer of forward_absorb
 *   (models/deepseek_v2.py:633-706, the ROCm-only fused form at :708; layers/layernorm.py:47-76;
 *    layers/rotary_embedding.py:143-169; mem_cache/memory_pool.py:439-452). */
int semipd_mla_decode_prep(void* q_nope_out, void* q_input, void* kv_buf, const float* planes, int n_planes,
                           int64_t plane_elems, const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                           const void* norm_weight, float eps, int64_t num_tokens, int num_q_heads, int nope_dim, int rope_dim,
                           int lora_rank, int64_t q_input_token_stride, int64_t q_input_head_stride, int64_t kvbuf_stride,
                           int dtype, int kv_dtype, void* stream);

/* The same launch for rows that are already tensors (the q_lora / block-fp8 path: q [num_tokens, Hq, nope + rope] out of
 * q_b_proj, latent [num_tokens, lora + rope] out of kv_a_proj_with_mqa, through their strides): kv_a_layernorm on the latent,
 * RoPE on q_pe and k_pe, q_pe -> q_input[t, h, lora : lora + rope], the finished latent row -> kv_buf row loc[t].  q and
 * latent are not modified.  Same bits as semipd_rmsnorm + semipd_rope_inplace_strided + the q_pe copy + semipd_kv_store_cvt.
 * replaces the same four statements of forward_absorb (models/deepseek_v2.py:670-684). */
int semipd_mla_decode_prep_rows(void* q_input, void* kv_buf, const void* q, const void* latent, const int64_t* loc,
                                const float* cos_sin_cache, const int64_t* positions, const void* norm_weight, float eps,
                                int64_t num_tokens, int num_q_heads, int nope_dim, int rope_dim, int lora_rank,
                                int64_t q_token_stride, int64_t q_head_stride, int64_t latent_token_stride,
                                int64_t q_input_token_stride, int64_t q_input_head_stride, int64_t kvbuf_stride, int dtype,
                                int kv_dtype, void* stream);

/* semipd_rope_kv_store for a decode batch whose qkv row is still the K-slice planes of semipd_stream_linear_planes:
 * planes [n_planes][num_tokens][(Hq + 2 Hk) * head] fp32 (plane stride plane_elems) are summed in slice order and
 * rounded to dtype -- the bits the GEMM's own reduction writes -- then q is rotated into q_out [num_tokens, q_stride],
 * k rotated into k_buf[loc[t]], v copied into v_buf[loc[t]].  neox pairing over the whole head (rot_dim == head_size).
 * One launch instead of the reduction + semipd_rope_kv_store; same bits.
 * replaces QKVParallelLinear.forward's output write + RotaryEmbedding.forward_cuda + set_kv_buffer for decode batches
 *   (layers/linear.py:165-172; layers/rotary_embedding.py:143-169; mem_cache/memory_pool.py:316-346). */
int semipd_rope_kv_store_planes(void* q_out, const float* planes, int n_planes, int64_t plane_elems, void* k_buf,
                                void* v_buf, const int64_t* loc, const float* cos_sin_cache, const int64_t* positions,
                                int64_t num_tokens, int num_q_heads, int num_k_heads, int head_size, int64_t q_stride,
                                int64_t kbuf_stride, int64_t vbuf_stride, int dtype, int kv_dtype, void* stream);

/* buf[loc[t], :row_elems] = src[t, :row_elems]   (byte-exact row scatter)
 * replaces MHATokenToKVPool.set_kv_buffer / MLATokenToKVPool.set_kv_buffer
 *   (mem_cache/memory_pool.py:316-346, 439-452). loc is int64 (out_cache_loc). */
int semipd_kv_store(void* buf, const void* src, const int64_t* loc, int64_t num_tokens,
                    int64_t row_bytes, int64_t buf_stride_bytes, int64_t src_stride_bytes,
                    void* stream);
/* Same scatter into an fp8 pool: buf[loc[t], :row_elems] = fp8(src[t, :row_elems]); strides in elements.
 * replaces `cache_k.to(self.dtype)` + index_put of set_kv_buffer (mem_cache/memory_pool.py:326-346). */
int semipd_kv_store_cvt(void* buf, const void* src, const int64_t* loc, int64_t num_tokens, int64_t row_elems,
                        int64_t buf_stride, int64_t src_stride, int dtype, int kv_dtype, void* stream);

/* ------------------------------------------------------------------ */
/* a4  kv_indptr / kv_indices                                          */
/* ------------------------------------------------------------------ */
/* kv_indptr[0]=0, kv_indptr[b+1]=kv_indptr[b]+lens[b]; kv_indices[kv_indptr[b]+j] =
 * req_to_token[req_pool_indices[b], start[b]+j], j<lens[b].  start may be NULL.
 * replaces create_flashinfer_kv_indices_triton + the cumsum around it
 *   (layers/attention/utils.py:5-39; triton_backend.py:76-200).
 * req_pool_indices int64 [B]; lens int64 [B] (seq_lens) or int32 selected by lens_is_i64. */
int semipd_build_kv_indices(const int32_t* req_to_token, int64_t req_to_token_stride,
                            const int64_t* req_pool_indices, const void* lens, int lens_is_i64,
                            const int32_t* start, int32_t* kv_indptr, int32_t* kv_indices,
                            int64_t batch, void* stream);

/* positions / extend_start_loc for an extend batch:
 * positions[start_loc[b]+j] = prefix_lens[b]+j  (j < extend_lens[b]),
 * start_loc = exclusive cumsum(extend_lens).
 * replaces compute_position_triton (model_executor/forward_batch_info.py:393-466). */
int semipd_compute_positions(const int32_t* prefix_lens, const int32_t* extend_lens,
                             int64_t* positions, int32_t* extend_start_loc, int64_t batch,
                             void* stream);

/* ------------------------------------------------------------------ */
/* a5  paged decode attention (flash-decoding, split-KV)               */
/* ------------------------------------------------------------------ */
/* q [B,Hq,Dk] (q_stride = elements between requests), K pool rows [N,Hkv,Dk]
 * (kbuf_stride = elements per pool row), V pool rows [N,Hkv,Dv]; kv_indptr
 * int32 [B+1]; kv_indices int32; attn_logits fp32 [B,Hq,num_kv_splits,Dv+1]
 * scratch (last lane = LSE); out [B,Hq,Dv].  logit_cap<=0 disables the cap.
 * For MLA pass Hkv=1, Dk=576, Dv=512 and v_buf = k_buf (latent rows).
 * replaces decode_attention_fwd
 *   (layers/attention/triton_ops/decode_attention.py:625-670: stage 1 :46-167 /
 *    :234-390, stage 2 :476-531). */
int semipd_decode_attention(void* out, const void* q, const void* k_buf, const void* v_buf,
                            const int32_t* kv_indptr, const int32_t* kv_indices,
                            float* attn_logits, int64_t batch, int num_q_heads, int num_kv_heads,
                            int head_dim_k, int head_dim_v, int64_t q_stride, int64_t o_stride,
                            int64_t kbuf_stride, int64_t vbuf_stride, int num_kv_splits,
                            float sm_scale, float logit_cap, int dtype, int kv_dtype, void* stream);

/* Everything between the qkv GEMM and o_proj of a decode step in ONE launch (GQA / MQA heads, neox RoPE over the whole
 * head): semipd_rope_kv_store_planes + semipd_decode_attention (both stages) with `waves` kv splits, same arithmetic in
 * the same order, same bits.  planes [n_planes][batch][(Hq + 2 Hkv) * head] fp32 = the K-slice planes of
 * semipd_stream_linear_planes; the rotated k and v of request b go to pool row loc[b], which must be the LAST row of the
 * request in kv_indices (the decode batch's new token); out [batch, o_stride] holds the merged attention output.
 * A workgroup = one (request, kv head) with 2 .. 16 q heads, its `waves` (4 or 8) waves are the kv splits and merge in
 * LDS: no attn_logits scratch, no stage-2 launch.  zsplits > 1 (batches that do not fill the chip with one workgroup per
 * pair): zsplits workgroups per (request, kv head), waves * zsplits kv splits in all; each workgroup merges its own and
 * writes ONE stage-1 partial to attn_logits fp32 [batch, Hq, zsplits, head + 1], and the stage-2 launch of
 * semipd_decode_attention merges zsplits partials per head (two launches instead of three, tolerance instead of bits).
 * _supported() tells whether the shape is instantiated (1) or the caller must use the separate calls (0).
 * replaces QKVParallelLinear's output write + RotaryEmbedding.forward_cuda + set_kv_buffer + decode_attention_fwd for
 *   decode batches (layers/linear.py:165-172; layers/rotary_embedding.py:143-169; mem_cache/memory_pool.py:316-346;
 *   layers/attention/triton_ops/decode_attention.py:625-670 with stage 1 :234-390 and stage 2 :476-531). */
int semipd_decode_rope_attention_planes_supported(int num_q_heads, int num_kv_heads, int head_size, int dtype, int kv_dtype);
int semipd_decode_rope_attention_planes(void* out, float* attn_logits, const float* planes, int n_planes, int64_t plane_elems,
                                        void* k_buf, void* v_buf, const int64_t* loc, const float* cos_sin_cache,
                                        const int64_t* positions, const int32_t* kv_indptr, const int32_t* kv_indices,
                                        int64_t batch, int num_q_heads, int num_kv_heads, int head_size, int64_t o_stride,
                                        int64_t kbuf_stride, int64_t vbuf_stride, int waves, int zsplits, float sm_scale,
                                        float logit_cap, int dtype, int kv_dtype, void* stream);


/* ------------------------------------------------------------------ */
/* a6  batched prefill (extend) attention                              */
/* ------------------------------------------------------------------ */
/* q_extend [T,Hq,Dk], k_extend [T,Hkv,Dk], v_extend [T,Hkv,Dv] contiguous rows
 * (strides in elements per token); paged prefix through kv_indptr/kv_indices
 * into k_buf/v_buf; qo_indptr int32 [B+1]; causal inside the extend part;
 * out [T,Hq,Dv].  replaces extend_attention_fwd
 *   (layers/attention/triton_ops/extend_attention.py:41-288, 291-410);
 * custom_mask == None of that call; semipd_extend_attention_masked below is the
 * call with a custom mask. */
int semipd_extend_attention(void* out, const void* q_extend, const void* k_extend,
                            const void* v_extend, const void* k_buf, const void* v_buf,
                            const int32_t* qo_indptr, const int32_t* kv_indptr,
                            const int32_t* kv_indices, int64_t batch, int num_q_heads,
                            int num_kv_heads, int head_dim_k, int head_dim_v, int64_t q_stride,
                            int64_t k_stride, int64_t v_stride, int64_t o_stride,
                            int64_t kbuf_stride, int64_t vbuf_stride, int max_len_extend,
                            float sm_scale, float logit_cap, int dtype, int kv_dtype,
                            void* stream);

/* extend_attention_fwd with its custom_mask / mask_indptr / skip_prefix_custom_mask
 * arguments (extend_attention.py:291-307; kernel :91-92, :164-177, :233-252): the
 * target-verify step of speculative decoding (triton_backend.py:136-149; off in
 * Semi-PD mode, managers/scheduler.py:272-275, so no caller inside this package).
 * custom_mask: one byte (torch.bool) per (new token, kv position), sequence b's
 * [len_extend_b][len_prefix_b + len_extend_b] block starting at mask_indptr[b]
 * (int64 [B+1]); a zero byte removes the key for that query.  With
 * skip_prefix_custom_mask != 0 (the reference default) the prefix columns are
 * not read: every prefix key is visible.  In the extend part the mask is AND-ed
 * with the causal triangle (the reference walks the keys up to the end of the
 * query's BLOCK_M tile there, so a bit above the diagonal counts or not with
 * the tile size of its launch; tree masks are sub-causal and unaffected).
 * A query whose every key is masked gets NaN, as in the reference. */
int semipd_extend_attention_masked(void* out, const void* q_extend, const void* k_extend,
                                   const void* v_extend, const void* k_buf, const void* v_buf,
                                   const int32_t* qo_indptr, const int32_t* kv_indptr,
                                   const int32_t* kv_indices, const uint8_t* custom_mask,
                                   const int64_t* mask_indptr, int skip_prefix_custom_mask,
                                   int64_t batch, int num_q_heads, int num_kv_heads, int head_dim_k,
                                   int head_dim_v, int64_t q_stride, int64_t k_stride,
                                   int64_t v_stride, int64_t o_stride, int64_t kbuf_stride,
                                   int64_t vbuf_stride, int max_len_extend, float sm_scale,
                                   float logit_cap, int dtype, int kv_dtype, void* stream);

/* ------------------------------------------------------------------ */
/* a8/a9  logits post-processing and greedy sampling                   */
/* ------------------------------------------------------------------ */
/* rows_out[b,:] = hidden[last_index[b],:]  — last-token gather
 * (layers/logits_processor.py:232-260). */
int semipd_gather_rows(void* out, const void* in, const int64_t* index, int64_t num_rows,
                       int64_t row_bytes, int64_t in_stride_bytes, void* stream);

/* next_token[b] = argmax_v logits[b,v] (lowest index wins ties); logits fp32
 * or bf16/f16 (upcast), out int32 or int64 selected by out_is_i64.
 * replaces Sampler greedy branch (layers/sampler.py:72-74). */
int semipd_argmax(const void* logits, void* out, int64_t batch, int64_t vocab,
                  int64_t logits_stride, int dtype, int out_is_i64, void* stream);

/* Fused lm_head + greedy: out[b] = argmax_v (hidden[b,:] . W[v,:]) computed with
 * bf16 MFMA / fp32 accumulate, optionally also writing fp32 logits [B,V]
 * (logits may be NULL).  W [V,H] row-major (lm_head.weight).
 * replaces LogitsProcessor._get_logits + Sampler argmax
 *   (layers/logits_processor.py:394-445; layers/sampler.py:72-74). */
int semipd_lm_head_argmax(const void* hidden, const void* weight, float* logits, void* out,
                          void* workspace, int64_t batch, int64_t hidden_size, int64_t vocab,
                          int dtype, int out_is_i64, void* stream);
size_t semipd_lm_head_argmax_workspace(int64_t batch, int64_t vocab);

/* Dense layer at decode batch sizes: out[rows, n] = x[rows, k] . weight[n, k]^T (bf16/f16, fp32
 * accumulate), rows <= 256, written for the regime where the call is bound by streaming `weight` once
 * from HBM (weight rows go straight into MFMA operand registers; split-K over `num_cus` compute units
 * with fp32 partial planes summed in a fixed order by a second small launch).  `workspace` (up to
 * semipd_linear_workspace bytes, reused by every call on one stream) enables split-K; NULL = no split.
 * num_cus = compute units this process may use (its CU-mask share), 0 = whole device.
 * replaces UnquantizedLinearMethod.apply -> F.linear for decode batches (layers/linear.py:165-172). */
size_t semipd_linear_workspace(int64_t max_rows, int64_t max_n);
int semipd_linear(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes,
                  int64_t rows, int64_t n, int64_t k, int64_t ldx, int64_t ldo, int num_cus, int dtype,
                  void* stream);

/* ---- tall decode batches (65 rows and up) and vocabulary-sized heads: the tiled ping-pong GEMM (csrc/gemm8p.hip) ------
 * out[rows, n_out] = x[rows, k] @ weight[n, k]^T; fuse_silu_mul: weight = merged [gate; up], n_out = n / 2 and the result
 * is SiluAndMul of the product rounded to dtype (the bits of the unfused pair).  256 x 256 output tiles, K in steps of
 * 64, both operands through LDS-DMA; workspace (optional, 16-byte aligned): fp32 planes for a K split when the tiles
 * alone do not fill the share declared with semipd_gemm_tall_set_cus (0 = 256).  Replaces F.linear (+ SiluAndMul) of
 * UnquantizedLinearMethod.apply for decode batches above the streaming kernel's 64 rows
 * (python/sglang/srt/layers/linear.py:165-172, models/llama.py:88-92, layers/activation.py:41-53) and the logits GEMM
 * of _get_logits (layers/logits_processor.py:394-445).  k % 64 == 0, n_out % 16 == 0, bf16 / f16.
 * semipd_gemm_tall_set_form(0 | 4 | 8): which of the two kernels runs the 256 x 256 tiles -- the 8-wave ping-pong kernel
 * or the 4-wave one (one 128 x 128 quarter of the tile per wave, a third less LDS traffic per K step; round 6); 0 (the
 * default) = four waves for the plain / planes epilogue, eight for SiLU * mul.  The same products, summed in the same
 * order per output element: the two forms agree bit for bit without a K split. */
int semipd_gemm_tall_set_cus(int cus);
int semipd_gemm_tall_set_form(int waves);
int semipd_gemm_tall(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes, int64_t rows,
                     int64_t n, int64_t k, int64_t ldx, int64_t ldo, int fuse_silu_mul, int dtype, void* stream);
/* The same GEMM (plain epilogue) stopped before the reduction over its K slices, for a row-parallel layer whose result
 * goes straight into RMSNorm(x, residual) (o_proj / down_proj of a prefill batch: python/sglang/srt/models/llama.py:279-302,
 * layers/linear.py:1258-1268 with tp_size == 1).  The split is the one semipd_gemm_tall would pick.  *ksplit > 1: fp32
 * planes [*ksplit][rows][n] in `planes`, `out` untouched; the consumer (semipd_fused_add_rmsnorm_planes) sums them in
 * slice order and rounds to dtype: the bits of semipd_gemm_tall + semipd_fused_add_rmsnorm.  *ksplit == 1: `out` holds
 * the result and `planes` is untouched. */
int semipd_gemm_tall_planes(void* out, float* planes, size_t planes_bytes, const void* x, const void* weight, int64_t rows,
                            int64_t n, int64_t k, int64_t ldx, int64_t ldo, int dtype, int* ksplit, void* stream);

/* The grouped form of the LDS-DMA streaming kernel for the expert GEMMs of DECODE-sized fused-MoE calls
 * (invoke_fused_moe_kernel, python/sglang/srt/layers/moe/fused_moe_triton/fused_moe.py:501-612): every touched expert's
 * weights are streamed once by row-shaped LDS-DMA.  sorted_token_ids / expert_ids from semipd_moe_align_block_size with
 * block size block_m in {16, 32, 48, 64} (T tokens route at most T rows to one expert: block_m = 16 ceil(T / 16));
 * max_sorted = entries of sorted_token_ids, a multiple of block_m.  c[id, :] = a[id / top_k_div, :] @ w[expert]^T for
 * id < num_valid, times topk_weights[id] when mul_routed_weight; fuse_silu_mul: w[e] = merged [gate; up], output width
 * n / 2, c = SiLU(gate) * up of the products rounded to dtype.  k % 128 == 0, output width % 16 == 0. */
int semipd_moe_stream_gemm(void* c, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
                           const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                           int64_t k, int64_t max_sorted, int top_k_div, int mul_routed_weight, int fuse_silu_mul, int block_m,
                           int dtype, void* stream);

/* The grouped form of the tiled ping-pong GEMM for the fused-MoE expert GEMMs of prefill-sized calls
 * (invoke_fused_moe_kernel, python/sglang/srt/layers/moe/fused_moe_triton/fused_moe.py:501-612; kernel :54-273):
 * sorted_token_ids / expert_ids from semipd_moe_align_block_size with block size block_m = 256 (256 x 256 tiles) or 128
 * (128 x 512 tiles): one expert per tile; max_sorted = entries of sorted_token_ids, a multiple of block_m.  c[id, :] = a[id / top_k_div, :] @ w[expert]^T for every
 * routed entry id < num_valid, times topk_weights[id] when mul_routed_weight; fuse_silu_mul: w[e] = merged [gate; up]
 * ([n, k], output width n / 2) and c = SiLU(gate) * up of the products rounded to dtype (the bits of the unfused pair). */
int semipd_moe_gemm_tall(void* c, const void* a, const void* w, const float* topk_weights, const int32_t* sorted_token_ids,
                         const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n, int64_t k,
                         int64_t max_sorted, int top_k_div, int mul_routed_weight, int fuse_silu_mul, int block_m, int dtype,
                         void* stream);

/* ---- prefill-sized dense layers on a CU share (csrc/dense_gemm.hip) ------------------------------------------------
 * out[rows, n] = x[rows, k] @ weight[n, k]^T (+ bias[n]) through hipBLASLt with the solution that MEASURED fastest on
 * the compute units this process owns.  Replaces F.linear in UnquantizedLinearMethod.apply
 * (python/sglang/srt/layers/linear.py:165-172) for batches above the streaming kernel's range: the reference sets the
 * prefill / decode shares at entrypoints/engine.py:583-634 and leaves GEMM selection to the library, whose heuristic
 * assumes the whole device.
 *   semipd_dense_gemm_init   creates the library handle and its workspace (0 = 64 MB); start-up.
 *   semipd_dense_gemm_tune   times the library's solutions for weight shape (n, k) at rows[0..num_rows) on this process's
 *                            CUs: its first num_heuristics heuristic results (0 = 64) at every row count, plus every
 *                            solution it has at the first num_full_search row counts (max_solutions > 0 caps that; ~20 s
 *                            per row count).  Allocates and frees scratch operands: start-up only.
 *   semipd_dense_gemm        launches with the winner of the nearest tuned row count (the library's own choice when
 *                            nothing was tuned or the winner does not support this row count).  Never allocates or
 *                            synchronises; dtype bf16 / f16; ldx / ldo = row strides in elements.
 *   semipd_dense_gemm_report text table of the tuning results; returns the bytes needed.
 *   semipd_dense_gemm_set_cus the CU count of the stream the following _tune / _dense_gemm calls run on: results are
 *                            filed under it and looked up by it (an instance that moves between its masked share and
 *                            the whole chip keeps one table per count; an untuned count gets the library's choice). */
int semipd_dense_gemm_init(size_t workspace_bytes);
int semipd_dense_gemm_set_cus(int cus);
/* Load a table written by semipd_dense_gemm_report instead of timing again (start-up cache, keyed by the caller on
 * architecture, CU count and library version: semipd_dense_gemm_library_version); *loaded = entries taken.  Skipped: indices
 * this build does not know, solutions that do not support the problem, and indices whose kernel NAME here differs from the
 * one recorded in the line (a table from another build of the library). */
int semipd_dense_gemm_import(const char* text, int* loaded);
int semipd_dense_gemm_library_version(int* version);
int semipd_dense_gemm_tune(int64_t n, int64_t k, const int64_t* rows, int num_rows, int num_full_search, int dtype,
                           int num_heuristics, int max_solutions, void* stream);
int semipd_dense_gemm(void* out, const void* x, const void* weight, const void* bias, int64_t rows, int64_t n, int64_t k,
                      int64_t ldx, int64_t ldo, int dtype, void* stream);
size_t semipd_dense_gemm_report(char* buf, size_t len);

/* semipd_stream_linear with the fp32 accumulators stored unrounded, no K split: out_f32[rows, n] (contiguous rows).
 * The logits GEMM of a decode batch of at most 64 rows (python/sglang/srt/layers/logits_processor.py:394-445;
 * semipd_lm_head_argmax takes this path when k % 128 == 0 and n % 16 == 0). */
int semipd_stream_linear_f32(float* out, const void* x, const void* weight, int64_t rows, int64_t n, int64_t k, int64_t ldx,
                             int dtype, void* stream);

/* Dense layer of a decode batch: out[rows, n_out] = x[rows, k] . weight[n, k]^T, rows <= 128 (bf16 / f16, fp32
 * accumulate).  Weights are streamed once through LDS-DMA rings; the launch is (n / 128 row batches) x KS slices of K
 * (up to 64 rows: eight waves of 16 weight rows, rings of three blocks; 65 .. 128 rows: four waves of 2 x 16 weight rows,
 * rings of two blocks -- the activation block is 32 KB there).
 * KS is a function of the shape AND of the CU count declared with semipd_stream_linear_set_cus (whole rounds of that
 * many workgroups): the slices set the order of the fp32 partial sums, so two processes produce the same bits exactly
 * when they declare the same count -- every instance of the engine declares the DEVICE's CU count for that reason
 * (ModelRunner.set_owned_cus; --k-split-by-share opts out).  With KS > 1 the slices write fp32 planes [KS][rows][n]
 * into `workspace` and a second small launch sums them in slice order; `workspace` NULL = no K split.  fuse_silu_mul
 * != 0: `weight` is a merged [gate; up] matrix (rows [0, n/2) gate, [n/2, n) up) and out[rows, n/2] = SiLU(gate) * up
 * with both GEMM outputs rounded to the activation type first, i.e. the value the unfused pair of ops produces.
 * k % 128 == 0.
 * replaces UnquantizedLinearMethod.apply -> F.linear (layers/linear.py:165-172) and, fused, LlamaMLP's
 *   gate_up_proj + SiluAndMul (models/llama.py:88-92, layers/activation.py:41-53) at decode batch sizes. */
/* Compute units the K split of semipd_stream_linear / _planes is sized for (whole rounds of that many workgroups);
 * 0 = default (128, half a chip).  The split sets the order of the fp32 partial sums: processes that must produce
 * identical bits declare the same count (the engine: the device's CU count in every instance).
 * Replaces nothing in the reference (its shares are MPS percentages, entrypoints/engine.py:591-593, 632-634, and its
 * GEMMs do not know them). */
int semipd_stream_linear_set_cus(int cus);
size_t semipd_stream_linear_workspace(int64_t max_n);
int semipd_stream_linear(void* out, const void* x, const void* weight, void* workspace, size_t workspace_bytes,
                         int64_t rows, int64_t n, int64_t k, int64_t ldx, int64_t ldo, int fuse_silu_mul,
                         int dtype, void* stream);

/* semipd_stream_linear stopped before its reduction: fp32 planes [*ksplit][rows][n] for a consumer that sums them
 * in slice order itself, and semipd_fused_add_rmsnorm_planes, that consumer for the o_proj / down_proj outputs:
 * x = T(sum of planes) (the row the GEMM would have written), residual += x, out = RMSNorm(residual) * weight --
 * the bits of semipd_stream_linear followed by semipd_fused_add_rmsnorm, two launches instead of three.
 * replaces RowParallelLinear.forward + RMSNorm.forward(x, residual) at decode batch sizes
 *   (layers/linear.py:1241-1270, layers/layernorm.py:47-76, models/llama.py:237-253). */
int semipd_stream_linear_planes(float* planes, size_t planes_bytes, const void* x, const void* weight, int64_t rows,
                                int64_t n, int64_t k, int64_t ldx, int dtype, int* ksplit, void* stream);
int semipd_fused_add_rmsnorm_planes(void* out, void* residual, const void* weight, const float* planes, int n_planes,
                                    int64_t plane_elems, int64_t num_tokens, int64_t hidden, float eps, int dtype,
                                    void* stream);

/* out[b, m, n] = dtype(sum_k X[b, m, k] * W[b, n, k]) for the MLA weight absorption of an UNQUANTISED model's decode batches:
 * X [batch, M, K] and W [batch, N, K] with K contiguous (16-byte fragments straight into v_mfma_f32_16x16x32_{bf16,f16}),
 * out through (batch, row) strides so that it lands in its consumer's layout (q_input[T, H, 576], the o_proj input).
 * k % 32 == 0, 16-byte aligned rows, bf16 / f16.
 * replaces torch.bmm(q_nope.transpose(0, 1), self.w_kc) and torch.bmm(attn_output.transpose(0, 1), self.w_vc)
 *   (models/deepseek_v2.py:655-667, 690-700). */
int semipd_bmm_nk(void* out, const void* x, const void* w, int64_t batch, int64_t m, int64_t n, int64_t k,
                  int64_t x_batch_stride, int64_t x_row_stride, int64_t w_batch_stride, int64_t w_col_stride,
                  int64_t out_batch_stride, int64_t out_row_stride, int dtype, void* stream);

/* Per-tensor fp8 for the MLA weight absorption of a block-fp8 DeepSeek model.
 * semipd_input_to_float8: q = fp8(clamp(x * scale)), scale = T(fp8_max / amax(|x|)) over the WHOLE tensor,
 *   *scale_inv = 1 / scale; x is [batch, m, k] through (batch, row) strides (a transposed view is fine), q is
 *   contiguous [batch, m, k]; amax_workspace = SEMIPD_INPUT_TO_FLOAT8_WORKSPACE_BYTES of device memory (per-block
 *   partial maxima; contents need not be initialised and do not carry over between calls).
 *   replaces input_to_float8 (layers/quantization/fp8_utils.py:137-149, called at models/deepseek_v2.py:659-661, 690-692).
 * semipd_bmm_fp8: out[b, m, n] = sum_k A[b, m, k] * B[b, k, n] * a_scale * b_scale, A row-major [batch, m, k], B
 *   COLUMN-major (memory [batch, n, k], like the reference's w_kc / w_vc buffers), fp8 matrix cores, fp32
 *   accumulate, bf16 / f16 output through (batch, row) strides.  e4m3fn x e4m3fn, e4m3fn x e5m2, e5m2 x e4m3fn.
 *   replaces torch.ops.sgl_kernel.bmm_fp8 (sgl-kernel/csrc/torch_extension.cc:146-149, csrc/gemm/bmm_fp8.cu,
 *   python/sgl_kernel/gemm.py:66-82; models/deepseek_v2.py:662-665, 693-700). */
#define SEMIPD_INPUT_TO_FLOAT8_WORKSPACE_BYTES 4096
int semipd_input_to_float8(void* q, float* scale_inv, void* amax_workspace, const void* x, int64_t batch, int64_t m,
                           int64_t k, int64_t x_batch_stride, int64_t x_row_stride, int dtype, int f8_dtype,
                           void* stream);
int semipd_bmm_fp8(void* out, const void* a, const void* b, const float* a_scale, const float* b_scale, int64_t batch,
                   int64_t m, int64_t n, int64_t k, int64_t a_batch_stride, int64_t a_row_stride,
                   int64_t b_batch_stride, int64_t b_col_stride, int64_t out_batch_stride, int64_t out_row_stride,
                   int a_f8_dtype, int b_f8_dtype, int out_dtype, void* stream);

/* Stochastic branch of Sampler.forward (layers/sampler.py:77-136).  All rows fp32, contiguous
 * [batch, vocab]; per-row parameter arrays may be NULL, then the scalar *_val applies to every row.
 *
 * logits <- softmax(logits / temperature[b]) in place (sampler.py:78-81). */
int semipd_softmax_temperature(float* logits, const float* temperatures, int64_t batch, int64_t vocab,
                               void* stream);
/* Joint top-k / top-p rejection sampling with uniform_samples [rounds, batch]; out_ids int32 [batch],
 * success uint8 [batch] (may be NULL).  replaces torch.ops.sgl_kernel.top_k_top_p_sampling_from_probs
 * with filter_apply_order="joint" (sgl-kernel/csrc/torch_extension.cc:166-170;
 * python/sgl_kernel/sampling.py:106-165; pinned by sgl-kernel/tests/test_sampling.py:8-52). */
int semipd_top_k_top_p_sampling_from_probs(const float* probs, const float* uniform_samples,
                                           const int32_t* top_ks, int32_t top_k_val, const float* top_ps,
                                           float top_p_val, int32_t* out_ids, uint8_t* success, int64_t batch,
                                           int64_t vocab, int rounds, void* stream);
/* Sample among probs >= min_p[b] * max_v probs[b,v] with uniform_samples [batch].
 * replaces min_p_sampling_from_probs (torch_extension.cc:160-164; tests/test_sampling.py:112-141). */
int semipd_min_p_sampling_from_probs(const float* probs, const float* uniform_samples, const float* min_ps,
                                     float min_p_val, int32_t* out_ids, int64_t batch, int64_t vocab,
                                     void* stream);
/* out = probs restricted to the top-k entries (ties at the k-th value kept), renormalised.
 * replaces top_k_renorm_probs_wrapper (torch_extension.cc:156-158; tests/test_sampling.py:84-109). */
int semipd_top_k_renorm_prob(const float* probs, float* out, const int32_t* top_ks, int32_t top_k_val,
                             int64_t batch, int64_t vocab, void* stream);
/* out = probs restricted to the smallest set of largest entries with mass >= top_p, renormalised.
 * replaces top_p_renorm_probs (torch_extension.cc:152-154; tests/test_sampling.py:57-81). */
int semipd_top_p_renorm_prob(const float* probs, float* out, const float* top_ps, float top_p_val,
                             int64_t batch, int64_t vocab, void* stream);
/* out[b] = log_softmax(logits[b])[ids[b]] (fp32), lse[b] = logsumexp(logits[b]) (may be NULL).
 * replaces the return_logprob branch of Sampler.forward on greedy batches: log_softmax + gather
 *   (layers/sampler.py:74-75, 139-155). */
int semipd_token_logprobs(const float* logits, const int32_t* ids, float* out, float* lse, int64_t batch,
                          int64_t vocab, void* stream);

/* ------------------------------------------------------------------ */
/* a10/a11/a12  MoE                                                    */
/* ------------------------------------------------------------------ */
/* softmax + top-k (+renormalise): gating [T,E] (dtype) -> topk_weights fp32 [T,k],
 * topk_ids int32 [T,k].  replaces fused_topk / vllm topk_softmax
 *   (layers/moe/topk.py:23-75). */
int semipd_topk_softmax(const void* gating, float* topk_weights, int32_t* topk_ids,
                        int64_t num_tokens, int num_experts, int topk, int renormalize, int dtype,
                        void* stream);

/* Group-limited routing (DeepSeek-V2/V3).  scoring: 0 softmax, 1 sigmoid.
 * correction_bias (fp32 [E]) may be NULL (grouped_topk, topk.py:79-117);
 * when given: biased_grouped_topk (topk.py:121-160). */
int semipd_grouped_topk(const void* gating, const float* correction_bias, float* topk_weights,
                        int32_t* topk_ids, int64_t num_tokens, int num_experts, int topk,
                        int num_expert_group, int topk_group, int renormalize, int scoring,
                        int dtype, void* stream);

/* semipd_grouped_topk on router logits that are still the fp32 K-slice planes [n_planes][num_tokens][num_experts] of the
 * router GEMM (semipd_stream_linear_planes on MoEGate.weight): summed in slice order and rounded to dtype inside the
 * routing kernel -- the bits of the GEMM's own reduction, one launch less, and for decode batches the weight-streaming
 * GEMM instead of the library's (18 us at 256 experts x 7168).
 * replaces MoEGate.forward's F.linear output write + grouped_topk / biased_grouped_topk
 *   (models/deepseek_v2.py:100-160, layers/moe/topk.py:79-160). */
int semipd_grouped_topk_planes(const float* planes, int n_planes, int64_t plane_elems, const float* correction_bias,
                               float* topk_weights, int32_t* topk_ids, int64_t num_tokens, int num_experts, int topk,
                               int num_expert_group, int topk_group, int renormalize, int scoring, int dtype, void* stream);

/* Counting sort of the T*k expert ids padded per expert to block_size.
 * replaces torch.ops.sgl_kernel.moe_align_block_size
 *   (torch_extension.cc:115-118; csrc/moe/moe_align_kernel.cu:27-161).
 * sorted_token_ids must hold T*k + E*(block_size-1) int32 and is filled with the
 * sentinel T*k in padding slots; expert_ids holds ceil(that/block_size). */
int semipd_moe_align_block_size(const int32_t* topk_ids, int64_t numel, int num_experts,
                                int block_size, int32_t* sorted_token_ids, int32_t* expert_ids,
                                int32_t* num_tokens_post_pad, int32_t* cumsum_buffer,
                                int64_t max_sorted, void* stream);

/* Grouped GEMM over expert-sorted rows (bf16 MFMA, fp32 accumulate):
 *   C[sorted row r, :N] = A[token(r), :K] @ W[expert(block(r)), :N, :K]^T
 * token(r) = sorted_token_ids[r] / top_k_div (top_k_div=topk for GEMM1 where A is
 * [T,K]; 1 for GEMM2 where A is [T*k,K]); rows with id >= num_valid are skipped.
 * mul_routed_weight multiplies each output row by topk_weights[sorted id].
 * C is [T*k, N] indexed by the sorted id.  block_m (64 or 128) must equal the block_size used in
 * semipd_moe_align_block_size: 64 for decode batches, 128 for prefill chunks (one pass over an
 * expert's weights per block_m rows routed to it).
 * replaces fused_moe_kernel / invoke_fused_moe_kernel
 *   (layers/moe/fused_moe_triton/fused_moe.py:54-273, 501-612). */
int semipd_moe_grouped_gemm(void* c, const void* a, const void* w, const float* topk_weights,
                            const int32_t* sorted_token_ids, const int32_t* expert_ids,
                            const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                            int64_t k, int64_t max_sorted, int top_k_div, int mul_routed_weight,
                            int block_m, int dtype, void* stream);

/* GEMM1 of fused_experts with the activation in its epilogue, for prefill-sized calls (the tiled kernel):
 *   c[sorted id, :n/2] = silu(T(A W_gate^T)) * T(A W_up^T),   w = [E, n, k], gate rows first
 * -- the bits of semipd_moe_grouped_gemm followed by semipd_silu_and_mul (whose input is the GEMM's rounded output),
 * without the [T*k, n] intermediate.  semipd_moe_grouped_gemm_silu_supported says whether a call qualifies (block_m 128,
 * at least 2048 routed rows, k % 64 == 0, n % 64 == 0, bf16 / f16); other calls keep the two separate entry points.
 * replaces invoke_fused_moe_kernel + SiluAndMul inside fused_experts_impl (fused_moe.py:1085-1125). */
int semipd_moe_grouped_gemm_silu_supported(int64_t num_valid, int64_t n, int64_t k, int top_k_div, int block_m, int dtype);
int semipd_moe_grouped_gemm_silu(void* c, const void* a, const void* w, const int32_t* sorted_token_ids,
                                 const int32_t* expert_ids, const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n,
                                 int64_t k, int64_t max_sorted, int top_k_div, int block_m, int dtype, void* stream);

/* moe_sum continued in registers by the element-wise tail of DeepseekV2MoE.forward (python/sglang/srt/models/
 * deepseek_v2.py:139-160): out = T(T(T(sum_j in[t,j,:]) * scale) + addend[t,:]); scale only when apply_scale, addend may be
 * NULL.  Every intermediate is rounded to the activation type like the separate kernels round it. */
int semipd_moe_sum_scale_add(void* out, const void* in, const void* addend, int64_t num_tokens, int topk, int64_t hidden,
                             float scale, int apply_scale, int dtype, void* stream);
/* The same with the addend still in the fp32 K-slice planes [n_planes][num_tokens][hidden] of the GEMM that produces it
 * (semipd_stream_linear_planes: the shared experts' down_proj of a decode batch): summed in slice order and rounded to
 * dtype inside this launch -- that GEMM's own reduction -- then added.  Same bits, one launch less.  bf16 / f16.
 * replaces DeepseekV2MLP.down_proj's output write + the tail of DeepseekV2MoE.forward (models/deepseek_v2.py:139-160). */
int semipd_moe_sum_scale_add_planes(void* out, const void* in, const float* add_planes, int n_planes, int64_t plane_elems,
                                    int64_t num_tokens, int topk, int64_t hidden, float scale, int apply_scale, int dtype,
                                    void* stream);

/* out[t,:] = sum_j in[t,j,:]   (vllm moe_sum, fused_moe.py:1144-1148) */
int semipd_moe_sum(void* out, const void* in, int64_t num_tokens, int topk, int64_t hidden,
                   int dtype, void* stream);

/* ------------------------------------------------------------------ */
/* 8f-4  block-scaled fp8 (layers/quantization/fp8_kernel.py, fused_moe_triton/fused_moe.py:174-243) */
/* ------------------------------------------------------------------ */
/* fp8 here is OCP e4m3fn (max 448), the format of gfx950's matrix cores; the reference's HIP branch
 * uses e4m3fnuz / 224 for MI300 (fp8_kernel.py:191-194).
 *
 * q[r, g*G .. (g+1)*G) = fp8(clamp(x * (1 / s[r, g]), +-448)), s[r, g] = max(absmax, eps) / 448;
 * replaces per_token_group_quant_fp8 (fp8_kernel.py:75-115, 165-250; row-major scales).
 * x [num_rows, hidden] contiguous f32/bf16/f16, hidden % group_size == 0, group_size in {64,128,256,512}. */
int semipd_per_token_group_quant_fp8(void* q, float* s, const void* x, int64_t num_rows, int64_t hidden,
                                     int group_size, float eps, int dtype, void* stream);
/* fused_add_rmsnorm (above) and per_token_group_quant_fp8 of the normalised rows in one pass: the activation
 * in front of a block-fp8 layer is quantised by the kernel that produces it.  inout / residual as in
 * semipd_fused_add_rmsnorm; q [num_tokens, hidden] fp8, qs [num_tokens, hidden / group_size]; hidden a multiple
 * of group_size, at most 8192; bf16 / f16.  Bytes and scales equal those of the two separate calls. */
int semipd_fused_add_rmsnorm_quant_fp8(void* inout, void* residual, const void* weight, void* q, float* qs,
                                       int64_t num_tokens, int64_t hidden, float eps, int group_size, float q_eps,
                                       int dtype, void* stream);
/* RMSNorm (no residual) of rows with a row stride, and the per-token-group fp8 quantisation of the result for the block-fp8
 * layer behind it, in one kernel: out [num_tokens, hidden] (row stride out_stride), q [num_tokens, hidden] e4m3fn dense,
 * qs [num_tokens, hidden / group_size].  The bytes of semipd_rmsnorm followed by semipd_per_token_group_quant_fp8.
 * replaces q_a_layernorm + the quantisation in front of q_b_proj (models/deepseek_v2.py:640-650; fp8_utils.py:91-134). */
int semipd_rmsnorm_quant_fp8(void* out, const void* input, const void* weight, void* q, float* qs, int64_t num_tokens,
                             int64_t hidden, int64_t in_stride, int64_t out_stride, float eps, int group_size, float q_eps,
                             int dtype, void* stream);
/* SiluAndMul (layers/activation.py:41-44) and per_token_group_quant_fp8 of its output in one pass, as
 * fused_experts_impl runs them back to back between its two GEMMs (fused_moe.py:1104-1125):
 * x [num_rows, 2*d] (gate | up) bf16/f16 -> q [num_rows, d] fp8, s [num_rows, d / group_size].  Bytes and
 * scales equal those of the two separate calls. */
int semipd_silu_and_mul_quant_fp8(void* q, float* s, const void* x, int64_t num_rows, int64_t d, int group_size,
                                  float eps, int dtype, void* stream);
/* c[m, n] = sum_kb (sum_{k in kb} a_q[m,k] * w_q[n,k]) * a_s[m,kb] * w_s[n / block_n, kb], fp32
 * accumulation; replaces w8a8_block_fp8_matmul (fp8_kernel.py:409-491, 694-800).
 * a_q [m,k] fp8, a_s [m, ceil(k/128)] f32, w_q [n,k] fp8, w_s [ceil(n/block_n), ceil(k/128)] f32, all
 * contiguous; block_k == 128, block_n % 16 == 0, k % 16 == 0; c [m,n] of out_dtype (f32/bf16/f16).
 * workspace (optional, 16-byte aligned device memory): room for fp32 split-K partials; calls with few
 * output tiles split K over up to 16 workgroups per tile when it is given. */
int semipd_w8a8_block_fp8_matmul(void* c, const void* a_q, const float* a_s, const void* w_q, const float* w_s,
                                 int64_t m, int64_t n, int64_t k, int block_n, int block_k, int out_dtype,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* semipd_moe_grouped_gemm with use_fp8_w8a8 and block_shape = [block_n, block_k]
 * (fused_moe.py:174-243, 516-545): a_q [num_valid / top_k_div, k] fp8 with a_s [.., ceil(k/128)],
 * w_q [E, n, k] fp8 with w_s [E, ceil(n/block_n), ceil(k/128)]; the other arguments as in
 * semipd_moe_grouped_gemm. */
int semipd_moe_grouped_gemm_fp8(void* c, const void* a_q, const float* a_s, const void* w_q, const float* w_s,
                                const float* topk_weights, const int32_t* sorted_token_ids, const int32_t* expert_ids,
                                const int32_t* num_tokens_post_pad, int64_t num_valid, int64_t n, int64_t k,
                                int64_t max_sorted, int top_k_div, int mul_routed_weight, int block_m, int block_n,
                                int block_k, int out_dtype, void* stream);

/* ------------------------------------------------------------------ */
/* a14  IPC seam (semi-pd-ipc/ipc.cpp:60-97)                           */
/* ------------------------------------------------------------------ */
/* handle = hipIpcMemHandle of the allocation that contains dev_ptr (64 bytes),
 * *offset = dev_ptr - allocation base.  replaces GetIPCMemHandle (ipc.cpp:60-64)
 * plus the _share_cuda_ offset lookup (semi_pd/utils.py:66-76). */
int semipd_ipc_get_handle(const void* dev_ptr, uint8_t handle[64], uint64_t* offset);
/* HIP runtime / driver version numbers (hipRuntimeGetVersion, hipDriverGetVersion): logged next to the IPC size rule
 * of semipd_ipc_get_handle, which was measured on one runtime.  library plumbing (no reference counterpart). */
int semipd_runtime_version(int* runtime, int* driver);
/* A stream the caller owns (hipStreamCreateWithFlags, non-blocking) for graph capture, and the way out of a capture
 * that failed half way.  Measured on ROCm 7.0 (tools/probe_capture_abort.py): an invalidated capture cannot be ended --
 * hipStreamEndCapture returns hipErrorStreamCaptureInvalidated and the stream stays invalidated -- and while it exists
 * every synchronising call of the process, on any stream, fails with that error; destroying the stream clears it.
 * semipd_stream_abort_capture = end the capture, drop the graph, DESTROY the stream (the handle is dead afterwards),
 * clear the sticky error.  That is why the decode graph runner captures on its own stream and not on one of torch's
 * pooled streams.  library plumbing (the reference relies on torch.cuda.graph's exit path,
 * model_executor/cuda_graph_runner.py:300-330). */
int semipd_stream_create(int device, void** stream);
int semipd_stream_abort_capture(void* stream);
/* Reads the runtime's per-thread last error until it is hipSuccess (returns how many were pending).  After an aborted
 * capture the host framework's own clean-up (graph and allocator bookkeeping on the dead stream) leaves
 * hipErrorInvalidValue behind, which its next launch check would report as that launch's failure. */
int semipd_clear_last_error(void);

/* Open (or re-use: one mapping per handle per process, ref-counted) and return the
 * mapped allocation base.  replaces ConvertIPCMemHandleToTensor's open
 * (ipc.cpp:67-85). */
int semipd_ipc_open(const uint8_t handle[64], int device, void** base);
/* Drop one reference; unmaps at zero (the reference never closes). */
int semipd_ipc_close(void* base);
/* Number of live mappings in this process (diagnostics / tests). */
int semipd_ipc_num_open(void);
/* replaces GetDeviceSMCount (ipc.cpp:87-92): CU count of `device`; like the
 * reference it also makes `device` current. */
int semipd_device_cu_count(int device, int* num_cus);

/* ------------------------------------------------------------------ */
/* a17  TP all-reduce over peer-mapped memory (sgl-kernel/csrc/torch_extension_rocm.cc:25-55,
 *      csrc/allreduce/custom_all_reduce.hip, custom_all_reduce_hip.cuh)                      */
/* ------------------------------------------------------------------ */
/* Every rank owns one shared region: a signal block (semipd_ar_meta_size bytes, replaces meta_size
 * custom_all_reduce.hip:117) followed by four payload slots of max_bytes each (input and reduced,
 * double buffered).  semipd_ar_region_size(max_bytes) is the size to allocate. */
size_t semipd_ar_meta_size(void);
size_t semipd_ar_region_size(size_t max_bytes);
/* Uncached, zeroed device memory for a region (replaces allocate_meta_buffer,
 * custom_all_reduce.hip:156-172).  Export it with semipd_ipc_get_handle (replaces
 * get_meta_buffer_ipc_handle :146-154) and map the peers' regions with semipd_ipc_open. */
int semipd_ar_alloc_shared(size_t bytes, void** ptr);
int semipd_ar_free_shared(void* ptr);
/* regions[world]: every rank's region as mapped in THIS process (own allocation at [rank]).
 * replaces init_custom_ar (custom_all_reduce.hip:13-51); world in {2, 4, 6, 8} like the reference
 * (:18-26).  There is no register_buffer / register_graph_buffers step: the kernel stages its input
 * into the rank's own region, so any input pointer works, also under hipGraph capture. */
int semipd_ar_init(void* const* regions, size_t region_bytes, int rank, int world, void** comm);
/* Bound every flag wait of the calls launched from now on (0 = wait for ever, the default and what the
 * reference does).  A wait that gives up is counted and the call finishes with undefined payload:
 * start-up self-tests use this to find a broken peer mapping without hanging the GPU. */
int semipd_ar_set_timeout_ms(void* comm, uint32_t ms);
/* Number of waits that gave up so far (synchronous device read). */
int semipd_ar_timed_out(void* comm, uint32_t* count);
/* Largest payload (bytes) one call can reduce. */
int semipd_ar_max_bytes(void* comm, size_t* bytes);
/* out = sum over ranks of in, fp32 accumulation in rank order (bit-identical on all ranks);
 * replaces all_reduce_reg / all_reduce_unreg (custom_all_reduce.hip:59-110).  numel * element size
 * must be a multiple of 16 (custom_all_reduce.py:451-453) and at most semipd_ar_max_bytes; in/out
 * 16-byte aligned; out may alias in.  Every rank must issue the same sequence of calls.  Does not
 * synchronise; capturable.  dtype: SEMIPD_F32 / BF16 / F16. */
int semipd_ar_all_reduce(void* comm, const void* in, void* out, size_t numel, int dtype, void* stream);
/* out[r * bytes_per_rank ...] = rank r's `in`, through the same regions and flags (the reference
 * gathers the vocab-parallel logits with the NCCL communicator, parallel_state.py:438-489,
 * logits_processor.py:426-427; here the decode graph then holds no RCCL node at all).
 * bytes_per_rank: multiple of 16, at most semipd_ar_max_bytes; out holds world * bytes_per_rank. */
int semipd_ar_all_gather(void* comm, const void* in, void* out, size_t bytes_per_rank, void* stream);
/* Expert-parallel all-to-all over the same regions and call sequence (BASELINE config 5).  The reference
 * has none: its expert parallelism keeps every token on every rank and all-reduces the partial outputs
 * (layers/moe/ep_moe/layer.py:190) -- these two calls are what would replace that all-reduce; their meaning is fixed by
 * oracle/ops.py: ep_dispatch / ep_combine (an index permutation and the sum of moe_sum).
 * semipd_ep_dispatch: this rank contributes `tokens` rows of `row_bytes` bytes (x), their routed GLOBAL expert ids
 *   [tokens, top_k] and routing weights; expert e lives on rank e / experts_per_rank.  It receives one row per (token, j)
 *   entry of ANY rank routed to one of its experts, ordered by sender rank, token, j: recv_x [max_recv, row_bytes],
 *   recv_expert (local expert id), recv_weight; recv_count[0] = rows received (rows beyond max_recv are dropped: the caller
 *   checks recv_count <= max_recv; max_recv MUST be the same number on every rank -- the way back relies on it); send_within [tokens, top_k] = position of this rank's entry among its entries to the same
 *   destination; counts_all [world, world] = entries every sender sends to every destination.  Every rank must call it in the
 *   same sequence position; tokens may differ per rank (also 0).  Staging needs 256 + 3 * align256(4 tokens top_k) +
 *   tokens * row_bytes bytes <= semipd_ar_max_bytes.  Capturable; does not synchronise.
 * semipd_ep_combine: y = this rank's expert outputs for the rows it received, in the received order (already multiplied by
 *   their routing weight); out[t] = T(sum over j, in j order, fp32, of the row that belongs to (t, j)): tokens, topk_ids,
 *   send_within, counts_all as in / from the dispatch; max_recv = the dispatch's (the same on every rank): an entry whose
 *   position at its destination is >= max_recv was dropped there and contributes ZERO here (never a read behind the staged
 *   rows) -- a caller that can overflow checks recv_count after the dispatch.  256 + max_recv * 2 * hidden bytes <=
 *   semipd_ar_max_bytes. */
int semipd_ep_dispatch(void* comm, const void* x, const int32_t* topk_ids, const float* topk_weights, int64_t tokens, int top_k,
                       int64_t row_bytes, int experts_per_rank, void* recv_x, int32_t* recv_expert, float* recv_weight,
                       int64_t max_recv, int32_t* recv_count, int32_t* send_within, int32_t* counts_all, void* stream);
int semipd_ep_combine(void* comm, const void* y, const int32_t* recv_count, int64_t max_recv, const int32_t* topk_ids,
                      const int32_t* send_within, const int32_t* counts_all, void* out, int64_t tokens, int top_k, int64_t hidden,
                      int experts_per_rank, int dtype, void* stream);
/* tests: from now on every block of every kernel launched on `comm` marks buf[xcc * 256 + hardware CU id] = 1 (buf: 2048
 * uint32 of device memory the caller zeroed; NULL = off): where a collective launched on a CU-masked stream really ran.
 * No counterpart in the reference (its MPS percentage confines NCCL implicitly, entrypoints/engine.py:591-593). */
int semipd_ar_set_cu_trace(void* comm, uint32_t* buf);
/* replaces dispose (custom_all_reduce.hip:112-115); regions stay with their owners. */
int semipd_ar_dispose(void* comm);

/* ------------------------------------------------------------------ */
/* a16  CU-mask compute isolation (replaces CUDA_MPS_ACTIVE_THREAD_PERCENTAGE,
 *      entrypoints/engine.py:591-593, 632-634; semi_pd/utils.py:10-11)     */
/* ------------------------------------------------------------------ */
/* Fill mask words so that `percent` of the device's CUs are enabled, spread evenly over the XCDs AND over the shader
 * engines of every XCD: on a 256-CU device the share is a whole number of groups of 32 logical CUs (8 XCDs x 4 shader
 * engines; bit i = XCD i % 8, shader engine (i / 8) % 4), elsewhere of groups of 8.  The dispatcher deals a kernel's
 * workgroups to the shader engines round-robin, so a share runs at the pace of its smallest engine: 48 or 56 CUs stream
 * like 32, 208 like 192 (profiles/r05_hbm_probe_shader_engine_balance.txt).  from_top selects the complementary (upper)
 * range so a prefill / decode pair can be disjoint.  Returns the number of CUs enabled.  words = ceil(num_cus / 32). */
int semipd_cu_mask_fill(int num_cus, int percent, int from_top, uint32_t* mask, int words);
/* hipExtStreamCreateWithCUMask wrapper; *stream receives a hipStream_t. */
int semipd_stream_create_cu_mask(int device, const uint32_t* mask, int words, void** stream);
int semipd_stream_destroy(void* stream);
/* A non-blocking stream with a HIP priority (hipStreamCreateWithPriority): -1 high, 0 normal, 1 low, clamped to the
 * range of the device; *range receives {least, greatest} when not NULL.  The second isolation knob next to the CU shares:
 * a prefill instance on a LOW-priority queue yields workgroup slots to the decode instance's kernels as they free up,
 * whatever CUs the two share.  Stands where the reference relies on the MPS daemon's time slicing between the two
 * processes (entrypoints/engine.py:588-593, 632-634). */
int semipd_stream_create_with_priority(int device, int priority, void** stream, int* range);
/* Read back the mask of a stream (hipExtStreamGetCUMask). */
int semipd_stream_get_cu_mask(void* stream, uint32_t* mask, int words);

/* Share board: one 4 KB page of HOST memory that the prefill and the decode instance of a GPU both map (a file in the
 * engine's socket directory), 64 slots of one int64 each, every slot written by one instance and read by the other with
 * release / acquire ordering.  It carries what the reference's MPS percentages cannot express: whether the other
 * instance has work in flight RIGHT NOW, so that an instance may run a step on every CU while the other one idles
 * (the reference's shares overlap -- P 80 %, D 100 %, semi_pd/utils.py:10-11 -- and MPS time-shares what overlaps; CU masks
 * are hard partitions, so the work-conserving part has to be decided per step by the instances themselves).
 * create != 0: make and zero the file if it is missing.  No reference counterpart. */
int semipd_share_board_open(const char* path, int create, void** board);
int semipd_share_board_close(void* board);
int semipd_share_board_store(void* board, int slot, int64_t value);
/* atomic slot += delta; *result (may be NULL) receives the new value. */
int semipd_share_board_add(void* board, int slot, int64_t delta, int64_t* result);
int semipd_share_board_load(void* board, int slot, int64_t* value);

/* Test/diagnostic kernel: every workgroup records the XCC id and CU id it ran on
 * (out[2*wg], out[2*wg+1]) and spins for `spin_cycles`. */
int semipd_probe_cu_placement(int32_t* out, int num_workgroups, int64_t spin_cycles, void* stream);

/* Measurement aid: enqueue `count` empty kernels.  bench.py brackets them with the same pair of HIP
 * events it uses around a hot kernel to calibrate the dispatch + completion overhead the pair adds. */
int semipd_launch_noop(int count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMIPD_H_ */
