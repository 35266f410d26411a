/*
 * sglk -- MI355X (gfx950) native kernels for SGLang's RadixAttention + quantized-matmul decode path.
 *
 * C ABI of libsglk.so.  Plain pointers and sizes only; every device pointer is a HIP device address,
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every launcher returns 0 on
 * success or a negative SGLK_E* code and leaves a message readable through sglk_last_error().
 * No launcher allocates, frees or synchronises, so all of them may be captured into a HIP graph.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository kexinchu/sglang-security).
 */
#ifndef SGLK_H_
#define SGLK_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGLK_API __attribute__((visibility("default")))

/* dtype codes */
#define SGLK_F32 0
#define SGLK_F16 1
#define SGLK_BF16 2
#define SGLK_FP8_E4M3 3 /* OCP e4m3fn: the native fp8 of gfx950 */

/* fp8 KV cache: a pool whose elements are one byte each (python/sglang/srt/mem_cache/memory_pool.py:119-123 stores
 * them as uint8; the pool type is float8_e5m2fnuz on ROCm, float8_e5m2 / float8_e4m3fn on CUDA,
 * model_executor/model_runner.py:847-854).  The attention entry points that say so take
 * `dtype | (SGLK_KV_* << SGLK_KV_SHIFT)`: q and o are of type `dtype`, k_buffer / v_buffer hold the named format, the
 * pool strides count bytes, and every element is converted exactly to fp32 as `.to(q.dtype)` does in the reference's
 * kernels (no scale is applied on the read side, as in triton_ops/decode_attention.py). */
#define SGLK_KV_E4M3FN 1
#define SGLK_KV_E5M2 2
#define SGLK_KV_E5M2FNUZ 3
#define SGLK_KV_SHIFT 8

/* error codes */
#define SGLK_OK 0
#define SGLK_EINVAL (-1)  /* bad shape / alignment / dtype: the reference raises RuntimeError here */
#define SGLK_EHIP (-2)    /* a HIP runtime call failed */
#define SGLK_EUNSUP (-3)  /* valid request this build has no kernel for */
#define SGLK_ESTUCK (-4)  /* an all-reduce communicator whose peer never reached a barrier: see sglk_ar_stuck */

SGLK_API const char* sglk_last_error(void);
SGLK_API int sglk_version(void);
/* number of compute units of the current device (TritonAttnBackend.device_core_count,
 * python/sglang/srt/layers/attention/triton_backend.py:233) */
SGLK_API int sglk_device_cu_count(void);

/* ------------------------------------------------------------------ A0: attention metadata ---- */

/* kv_indices[kv_indptr[b] + j] = req_to_token[req_pool_indices[b], kv_start[b] + j], j < lens[b].
 * Replaces create_flashinfer_kv_indices_triton, python/sglang/srt/layers/attention/utils.py:5-41.
 * idx/lens/start may be int32 or int64 (the *_is64 flags); kv_start may be NULL. */
SGLK_API int sglk_create_kv_indices(const int32_t* req_to_token, int64_t row_stride,
                                    const void* req_pool_indices, int idx_is64, const void* lens,
                                    int lens_is64, const int32_t* kv_indptr, const void* kv_start,
                                    int start_is64, int32_t* kv_indices, int bs, void* stream);

/* kv_indptr[0] = 0, kv_indptr[1+b] = sum_{i<=b} seq_lens[i].
 * Replaces `kv_indptr[1:bs+1] = torch.cumsum(seq_lens)`, triton_backend.py:283. */
SGLK_API int sglk_kv_indptr_cumsum(const void* seq_lens, int lens_is64, int32_t* kv_indptr, int bs,
                                   void* stream);

/* Per-request split count. Replaces get_num_kv_splits_triton, triton_backend.py:23-72, 235-269. */
SGLK_API int sglk_get_num_kv_splits(int32_t* num_kv_splits, const void* seq_lens, int lens_is64,
                                    int num_seq, int num_group, int num_head, int num_kv_head,
                                    int max_kv_splits, int device_core_count, void* stream);

/* One launch for the three above (graph replay refreshes metadata at fixed addresses,
 * triton_backend.py init_forward_metadata_replay_cuda_graph). */
SGLK_API int sglk_decode_metadata(const int32_t* req_to_token, int64_t row_stride,
                                  const void* req_pool_indices, int idx_is64, const void* seq_lens,
                                  int lens_is64, int32_t* kv_indptr, int32_t* kv_indices,
                                  int32_t* num_kv_splits, int bs, int num_head, int num_kv_head,
                                  int max_kv_splits, int device_core_count, void* stream);

/* ------------------------------------------------------------------ A1: decode attention ------ */

/* o[b,h,:] = softmax(sm_scale * q[b,h] . K[idx(b)]^T  [logit_cap * tanh(./logit_cap)]) . V[idx(b)]
 * with GQA head mapping h -> h / (Hq/Hkv), split-KV partials in attn_logits [B,Hq,max_kv_splits,Dv]
 * fp32 and attn_lse [B,Hq,max_kv_splits] fp32, merged by LSE.
 * Replaces decode_attention_fwd, python/sglang/srt/layers/attention/triton_ops/decode_attention.py:682-733
 * (== decode_attention_cpu, sgl-kernel/csrc/cpu/decode.cpp:1064-1207, without its KV write).
 * Strides are in elements; the last dimension of every tensor is contiguous.
 * Matrix-core kernels serve 16-bit types with head size 128, head size 64 with an even number of contiguous kv heads
 * (k_stride_h == v_stride_h == 64: pairs of kv heads are walked as one 128-element row) and the absorbed-MLA shape
 * 576 / 512 with v_buffer a view of k_buffer; every other shape or stride takes a generic kernel.
 * fp8 pools: dtype | (SGLK_KV_* << SGLK_KV_SHIFT) with fp16 / bf16 queries, pool strides in bytes; head size 128 runs
 * the matrix-core kernel (8-byte row pieces widened to the query type through a 256-entry table on their way into
 * LDS), other shapes the generic kernel.  The same holds for the _xs and shared-prefix entry points below (an fp8
 * pool takes the per-request kernels there). */
SGLK_API int sglk_decode_attention(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                   const int32_t* kv_indptr, const int32_t* kv_indices,
                                   float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                                   int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv,
                                   int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n,
                                   int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                   int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
                                   float logit_cap, int dtype, void* stream);

/* Radix-prefix-shared decode: requests are partitioned into groups (group_indptr [G+1] over the
 * batch, requests of a group contiguous); the first prefix_lens[g] KV slots of every request in
 * group g are the same pool slots (a radix-tree hit), so that part of K/V is read once per group
 * and multiplied against all of the group's query heads on the matrix cores; the per-request
 * remainder is walked per request, and the two partial results are merged by LSE
 * (merge_state_v2 arithmetic, sgl-kernel/csrc/attention/merge_attn_states.cu:31-106).
 * Same result as sglk_decode_attention on the same inputs (fp tolerance); no counterpart kernel in
 * the reference, whose decode walks every request's full kv_indices
 * (triton_ops/decode_attention.py:107-150).
 * One launch: blocks that stream a slice of a shared prefix (lowest block indices, never waiting) leave fp32
 * partials in the workspace and count them per (request, kv head); the blocks that walk one request's
 * remainder then wait for that count, merge and write the output.  The wait is bounded (2 s): on expiry the
 * block writes NaN and raises the workspace's error word (sglk_decode_shared_workspace_status returns 1;
 * synchronising) instead of hanging the GPU.
 * workspace: sglk_decode_shared_workspace_bytes() bytes, 16-byte aligned.  Its first B * Hq + 1 int32 words
 * are counters: they must be ZERO when the workspace is first used (sglk_decode_shared_workspace_init clears
 * them; a zero-filled allocation does too) and every call leaves them zero.  Calls that overlap in time need
 * workspaces of their own.
 * The one-launch kernel of the group-array entry points covers 16-bit types with head size 128.  Every other shape
 * sglk_decode_attention accepts (absorbed MLA 576 / 512 with the values a view of the key rows, head size 64, fp8
 * pools, fp32) is routed there to the per-request kernels over the same lists with max_kv_splits splits per request,
 * their partials in the workspace: the same result, a group's shared rows fetched once per request (from L2 after the
 * first).  The planned entry point below covers more shapes on the one-launch kernel. */
/* The per-step plan of the shared-prefix kernel.  A decode step calls the kernel once per layer over the same
 * request table, so the groups, every request's remainder range and every prefix block's slot range, column owners
 * and first pool slots are resolved once per step, ON THE DEVICE, into `plan` (sglk_decode_shared_plan_bytes() bytes,
 * 16-byte aligned, contents rewritten by every plan call: the fixed-address metadata a captured graph replays over,
 * python/sglang/srt/model_executor/cuda_graph_runner.py:491-533).  B <= 1024.
 *   sglk_decode_shared_plan: groups are found from the table itself -- requests whose lists start with the same
 *     pool slot matched the same radix-tree node (mem_cache/radix_cache.py:120-150 hands every such request the same
 *     slots); a group's shared length is the longest slot prefix common to all its members, the newest token
 *     excluded.  Members need not be neighbours in the batch.  Groups of one request, and groups sharing fewer than
 *     min_share slots (0: 32), take no prefix blocks.  One level: requests that share a short system prompt and, in
 *     sub-groups, longer prefixes form ONE group with the short length.
 *   sglk_decode_shared_plan_from_groups: the caller's groups (group_indptr / prefix_lens as for
 *     sglk_decode_attention_shared_prefix, members contiguous), kept as given.
 *   sglk_decode_metadata_shared: sglk_decode_metadata followed by sglk_decode_shared_plan, one call for the
 *     backend's init_forward_metadata / ..._replay_cuda_graph (triton_backend.py:281-337,627-730).
 *   sglk_decode_attention_shared_planned: the kernel of sglk_decode_attention_shared_prefix_xs driven by a plan
 *     (same workspace, same result).  Its grid depends on B and the head counts only, never on the groups, and it
 *     serves every batch by itself: a request that shares nothing walks its whole context as its "remainder", over
 *     up to max_kv_splits / 4 blocks of four waves where the launch has room for them.  One-launch shapes: 16-bit
 *     queries with head size 128 over 16-bit or fp8 pools (dtype | SGLK_KV_* << 8), and head size 64 with an even
 *     number of kv heads whose rows are contiguous -- then the plan must have been laid out with num_kv_head / 2
 *     (pairs of kv heads are walked as one 128-wide head).  Absorbed MLA (576 / 512, one latent head) runs the plan's
 *     items on the MLA tile kernel plus a merge launch.  The kernel refuses a plan laid out for another batch size,
 *     head count or split count (NaN rows and the workspace's error word).  Other shapes: per-request kernels.
 *     num_kv_splits (may be NULL): the per-request split counts of sglk_decode_metadata; used by the absorbed-MLA form
 *     only, whose per-request kernels are launched beside the plan-driven ones, a word of the plan deciding ON THE
 *     DEVICE which of the two works (no group shares anything: per request) -- what a captured graph needs. */
SGLK_API int64_t sglk_decode_shared_plan_bytes(int B, int Hq, int Hkv, int max_kv_splits);
SGLK_API int sglk_decode_shared_plan(const int32_t* req_to_token, int64_t row_stride,
                                     const void* req_pool_indices, int idx_is64, const void* seq_lens,
                                     int lens_is64, const int32_t* kv_indptr, const int32_t* kv_indices, int B,
                                     int Hq, int Hkv, int max_kv_splits, int min_share, int32_t* plan,
                                     void* stream);
SGLK_API int sglk_decode_shared_plan_from_groups(const int32_t* kv_indptr, const int32_t* kv_indices,
                                                 const int32_t* group_indptr, const int32_t* prefix_lens, int G,
                                                 int B, int Hq, int Hkv, int max_kv_splits, int32_t* plan,
                                                 void* stream);
SGLK_API int sglk_decode_metadata_shared(const int32_t* req_to_token, int64_t row_stride,
                                         const void* req_pool_indices, int idx_is64, const void* seq_lens,
                                         int lens_is64, int32_t* kv_indptr, int32_t* kv_indices,
                                         int32_t* num_kv_splits, int bs, int num_head, int num_kv_head,
                                         int max_kv_splits, int device_core_count, int min_share, int32_t* plan,
                                         void* stream);
SGLK_API int sglk_decode_attention_shared_planned(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, const int32_t* plan, float* workspace, int max_kv_splits, int B, int Hq, int Hkv,
    int D, int Dv, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h,
    int64_t v_stride_n, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
    float logit_cap, int dtype, float* xsum_out, int64_t xs_stride, const int32_t* num_kv_splits, void* stream);
/* sglk_rope_and_store_kv_partial and sglk_decode_attention_shared_planned in ONE launch: the rotation + pool write
 * (models/llama.py:176-178 rotary_emb, then RadixAttention's set_kv_buffer, triton_backend.py:806-810) runs as the
 * first blocks of the attention launch; the attention blocks fetch their slot lists and K / V rows meanwhile and wait
 * per request, on a counter, only where they first need a query row or the new row's slot.  q_out receives the rotated
 * query rows (row stride q_stride_b elements; the attention reads them from there), the pools receive the new K / V
 * rows; same bits as the two separate entry points.  launch_in_step: 1 for a step's first call on this plan, 2 for
 * its second, ... (the counters live in the plan, which sglk_decode_shared_plan* clears once per step; a decoder
 * layer's index + 1).  16-bit types, D = Dv = 128, rot_dim = 128, interleave = 0; plan is written (the counters). */
SGLK_API int sglk_decode_attention_shared_planned_rope(
    const float* qkv_partials, int splits, int64_t part_stride, const void* qkv_bias, const float* cos_sin_cache,
    const int64_t* positions, int rot_dim, int interleave, const void* loc, int loc_is64, void* q_out,
    int64_t q_stride_b, void* k_buffer, void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, int32_t* plan, int launch_in_step, float* workspace, int max_kv_splits, int B,
    int Hq, int Hkv, int D, int Dv, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
    int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype,
    float* xsum_out, int64_t xs_stride, void* stream);
SGLK_API int64_t sglk_decode_shared_workspace_bytes(int B, int Hq, int Dv, int max_kv_splits);
SGLK_API int sglk_decode_shared_workspace_init(void* workspace, int B, int Hq, void* stream);
SGLK_API int sglk_decode_shared_workspace_status(const void* workspace, int B, int Hq);
SGLK_API int sglk_decode_attention_shared_prefix(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, const int32_t* group_indptr, const int32_t* prefix_lens, int G,
    float* workspace, int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, void* stream);

/* ScheduleBatch.prepare_for_decode on device-resident batch state (python/sglang/srt/managers/schedule_batch.py:
 * 1464-1527): positions[b] = seq_lens[b]; out_cache_loc[b] = req_to_token[req_pool_indices[b], seq_lens[b]] (the
 * slot the request table already holds for the next position); seq_lens[b] += 1.  One launch. */
SGLK_API int sglk_prepare_for_decode(const int32_t* req_to_token, int64_t row_stride,
                                     const void* req_pool_indices, int idx_is64, void* seq_lens,
                                     int lens_is64, int64_t* positions, void* out_cache_loc,
                                     int loc_is64, int bs, void* stream);
/* Rotary step of the absorbed-MLA decode with fused rotation (decode_attention_fwd_grouped_rope,
 * python/sglang/srt/layers/attention/triton_ops/rocm_mla_decode_rope.py:45-202,402-440): q_out = q with the rope
 * part (elements kv_lora_rank .. + rope_dim, the first rotary_dim of them rotated) turned at positions[b]; the rope
 * part of each request's newest pool row (kv_indices[kv_indptr[b + 1] - 1], which the caller stored unrotated) is
 * rotated IN PLACE in kv_buffer and copied to k_pe_out [B, rope_dim] -- the reference leaves the pool row as it is
 * and the caller then stores k_pe_out over it (models/deepseek_v2.py:1189-1193), after which the two agree.
 * Follow with sglk_decode_attention on q_out.  cos_sin_cache fp32 [max_pos, rotary_dim] (cos half, sin half). */
/* Rotary step + cache write of the absorbed-MLA decode in one launch (forward_absorb,
 * python/sglang/srt/models/deepseek_v2.py:1127-1193: rotary_emb on q_pe / k_pe, q_input[..., lora:] = q_pe,
 * k_input = [k_nope | k_pe], set_kv_buffer of the latent row): for request b the rope part of every head,
 * q_pe[b, h, 0..rope_dim) (strides qpe_stride_b / _h), is rotated at positions[b] into
 * q_out[b, h, kv_lora_rank .. + rope_dim); the latent row latent[b] = [k_nope (kv_lora_rank, already normalised) |
 * k_pe (rope_dim, unrotated)] is written to kv_buffer row loc[b] with its rope part rotated.  The first rotary_dim
 * rope elements are rotated (neox halves or interleaved pairs), the rest copied.  cos_sin_cache fp32
 * [max_pos, rotary_dim] (cos half, sin half). */
SGLK_API int sglk_mla_rope_and_store(const void* q_pe, void* q_out, const void* latent, void* kv_buffer,
                                     const void* loc, int loc_is64, const float* cos_sin_cache,
                                     const int64_t* positions, int B, int H, int kv_lora_rank, int rope_dim,
                                     int rotary_dim, int64_t qpe_stride_b, int64_t qpe_stride_h,
                                     int64_t qo_stride_b, int64_t qo_stride_h, int64_t latent_stride_b,
                                     int64_t kv_stride_n, int64_t cos_sin_stride, int is_neox, int dtype,
                                     void* stream);
SGLK_API int sglk_mla_rope_prepare(const void* q, void* q_out, void* kv_buffer, void* k_pe_out,
                                   const int32_t* kv_indptr, const int32_t* kv_indices,
                                   const float* cos_sin_cache, const int64_t* positions, int B, int H,
                                   int kv_lora_rank, int rope_dim, int rotary_dim, int64_t q_stride_b,
                                   int64_t q_stride_h, int64_t qo_stride_b, int64_t qo_stride_h,
                                   int64_t kv_stride_n, int64_t kpe_stride_b, int64_t cos_sin_stride,
                                   int is_neox, int dtype, void* stream);

/* Greedy sampling: out[r] = index of the first maximum of row r (torch.argmax(logits, -1),
 * python/sglang/srt/layers/sampler.py greedy branch). */
SGLK_API int sglk_argmax(const void* x, int64_t* out, int64_t rows, int cols, int64_t row_stride,
                         int dtype, void* stream);

/* Longest common prefix (in slots) of the batch's req_to_token rows, one group covering the whole
 * batch: out[0] = min over b of the first j with req_to_token[idx[b], j] != req_to_token[idx[0], j],
 * capped at min(seq_lens) - 1.  Device-side helper for the shared-prefix path when the scheduler
 * does not hand over radix-tree groups. */
SGLK_API int sglk_common_prefix_len(const int32_t* req_to_token, int64_t row_stride,
                                    const void* req_pool_indices, int idx_is64, const void* seq_lens,
                                    int lens_is64, int bs, int32_t* out, void* stream);

/* ------------------------------------------------------------------ A2: extend attention ------ */

/* Prefill with a cached prefix: stage 1 over the prefix through kv_indices, stage 2 causal over the
 * contiguous new tokens.  Replaces extend_attention_fwd,
 * python/sglang/srt/layers/attention/triton_ops/extend_attention.py:306-438
 * (== extend_attention_cpu, sgl-kernel/csrc/cpu/extend.cpp:481-621).  For a custom mask see
 * sglk_extend_attention_masked.
 * fp8 pools: dtype | (SGLK_KV_* << SGLK_KV_SHIFT): k_buffer / v_buffer hold one-byte codes (strides in bytes), the new
 * tokens' k_extend / v_extend stay in the query type.  Accepted by every extend entry point of this section
 * (masked, window, split, paged prefill). */
SGLK_API int sglk_extend_attention(const void* q_extend, const void* k_extend, const void* v_extend,
                                   void* o_extend, const void* k_buffer, const void* v_buffer,
                                   const int32_t* qo_indptr, const int32_t* kv_indptr,
                                   const int32_t* kv_indices, int B, int Hq, int Hkv, int D, int Dv,
                                   int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                   int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h,
                                   int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n,
                                   int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                   int max_extend_len, float sm_scale, float logit_cap, int causal,
                                   int dtype, void* stream);

/* sglk_extend_attention under a custom mask (draft-tree verification): custom_mask uint8, for request b the
 * [extend_len, prefix_len + extend_len] row-major block at mask_indptr[b] (int64 [B + 1]); a new token j is visible
 * to row i iff its byte is non-zero, prefix tokens are all visible when skip_prefix_custom_mask, else by their bytes
 * (extend_attention.py:93-94,167-178,245-257).  Mask bytes above the diagonal are honoured as given (the reference,
 * when is_causal, drops those beyond its row tile); a row that sees nothing is written as zeros (reference: NaN). */
SGLK_API int sglk_extend_attention_masked(const void* q_extend, const void* k_extend, const void* v_extend,
                                          void* o_extend, const void* k_buffer, const void* v_buffer,
                                          const int32_t* qo_indptr, const int32_t* kv_indptr,
                                          const int32_t* kv_indices, const uint8_t* custom_mask,
                                          const int64_t* mask_indptr, int skip_prefix_custom_mask, int B,
                                          int Hq, int Hkv, int D, int Dv, int64_t q_stride_t,
                                          int64_t q_stride_h, int64_t ke_stride_t, int64_t ke_stride_h,
                                          int64_t ve_stride_t, int64_t ve_stride_h, int64_t o_stride_t,
                                          int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                          int64_t v_stride_n, int64_t v_stride_h, int max_extend_len,
                                          float sm_scale, float logit_cap, int dtype, void* stream);

/* sglk_extend_attention with the cached prefix split over kv_splits blocks per (sequence, query block, head) and a
 * second launch that merges the partials by log-sum-exp: for batches with few (query block, head) pairs -- a single
 * request extending a short tail over a long cached prefix, the shape of a radix-cache hit -- where one block per pair
 * leaves most of the chip idle and walks the whole prefix serially.  No counterpart in the reference, whose kernel
 * walks the prefix once per query block (extend_attention.py:96-210); same result within fp tolerance (the merge is
 * merge_state arithmetic in split order).  16-bit types, head size 128, contiguous new tokens, no mask.
 * workspace: sglk_extend_attention_split_workspace_bytes(total_rows, Hq, kv_splits) bytes, 16-byte aligned, where
 * total_rows = qo_indptr[B] (the number of query rows); kv_splits == 1 is sglk_extend_attention. */
SGLK_API int64_t sglk_extend_attention_split_workspace_bytes(int64_t total_rows, int Hq, int kv_splits);
SGLK_API int sglk_extend_attention_split(const void* q_extend, const void* k_extend, const void* v_extend,
                                         void* o_extend, const void* k_buffer, const void* v_buffer,
                                         const int32_t* qo_indptr, const int32_t* kv_indptr,
                                         const int32_t* kv_indices, int B, int Hq, int Hkv, int D, int Dv,
                                         int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                         int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h,
                                         int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n,
                                         int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                         int max_extend_len, float sm_scale, float logit_cap, int causal,
                                         int kv_splits, float* workspace, int64_t total_rows, int dtype,
                                         void* stream);

/* sglk_extend_attention for a sliding-window layer (extend_attention.py:179-184,323 as called by
 * triton_backend.py:759-790).  The caller lists, as the reference's backend does (update_sliding_window_buffer,
 * triton_backend.py:75-103), only the last min(prefix_len, window + 1) prefix tokens of every request in kv_indices;
 * in the stage over that list row i of the request's extend sees listed token n iff i <= n + sliding_window_size
 * (both counted from zero: the reference's formula, kept as it stands); the stage over the new tokens is causal (or
 * custom_mask, if given: same meaning as in sglk_extend_attention_masked, its prefix columns being the listed ones)
 * without a window term, as in the reference.  sliding_window_size <= 0: no window.  Served by the wave-per-row
 * kernel. */
SGLK_API int sglk_extend_attention_window(const void* q_extend, const void* k_extend, const void* v_extend,
                                          void* o_extend, const void* k_buffer, const void* v_buffer,
                                          const int32_t* qo_indptr, const int32_t* kv_indptr,
                                          const int32_t* kv_indices, const uint8_t* custom_mask,
                                          const int64_t* mask_indptr, int skip_prefix_custom_mask,
                                          int sliding_window_size, int B, int Hq, int Hkv, int D, int Dv,
                                          int64_t q_stride_t, int64_t q_stride_h, int64_t ke_stride_t,
                                          int64_t ke_stride_h, int64_t ve_stride_t, int64_t ve_stride_h,
                                          int64_t o_stride_t, int64_t o_stride_h, int64_t k_stride_n,
                                          int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                          int max_extend_len, float sm_scale, float logit_cap, int causal,
                                          int dtype, void* stream);

/* Paged prefill: like sglk_extend_attention, but the new tokens are already in the pool and kv_indices lists
 * every token of each sequence (prefix + new); with causal != 0 query row i of a sequence with q_len rows and
 * kv_len tokens sees tokens 0 .. kv_len - q_len + i.  This is the form the reference's ROCm default backend
 * calls (aiter.mha_batch_prefill_func, python/sglang/srt/layers/attention/aiter_backend.py:353-367). */
SGLK_API int sglk_prefill_attention_paged(const void* q, void* o, const void* k_buffer, const void* v_buffer,
                                          const int32_t* qo_indptr, const int32_t* kv_indptr,
                                          const int32_t* kv_indices, int B, int Hq, int Hkv, int D, int Dv,
                                          int64_t q_stride_t, int64_t q_stride_h, int64_t o_stride_t,
                                          int64_t o_stride_h, int64_t k_stride_n, int64_t k_stride_h,
                                          int64_t v_stride_n, int64_t v_stride_h, int max_q_len, float sm_scale,
                                          float logit_cap, int causal, int dtype, void* stream);

/* ------------------------------------------------------------------ A3: merge_state ----------- */

/* LSE merge of two partial attention results, v:[n,h,d] (dtype), s:[n,h] fp32; s_out may be NULL.
 * Replaces merge_state_v2, sgl-kernel/csrc/attention/merge_attn_states.cu:182-204
 * (python/sgl_kernel/attention.py:31-51). */
SGLK_API int sglk_merge_state(const void* v_a, const float* s_a, const void* v_b, const float* s_b,
                              void* v_out, float* s_out, int64_t n, int h, int d, int dtype,
                              void* stream);

/* ------------------------------------------------------------------ P1: KV pool write --------- */

/* k_buffer[loc[t]] = k[t], v_buffer[loc[t]] = v[t]  (rows of Hkv*D / Hkv*Dv elements).
 * Replaces MHATokenToKVPool.set_kv_buffer, python/sglang/srt/mem_cache/memory_pool.py:378-412. */
SGLK_API int sglk_set_kv_buffer(void* k_buffer, void* v_buffer, const void* k, const void* v,
                                const void* loc, int loc_is64, int64_t tokens, int row_k, int row_v,
                                int64_t kbuf_stride_n, int64_t vbuf_stride_n, int64_t k_stride_t,
                                int64_t v_stride_t, int elem_bytes, void* stream);

/* The same write into an fp8 pool (format SGLK_KV_*): byte = fp8(k[t] / k_scale), rounded as torch's
 * Tensor.to(float8 type) rounds (nearest even, denormals kept; out of range: NaN for e4m3fn and e5m2fnuz, infinity for
 * e5m2; the fnuz format has no negative zero).  k / v are rows of `src_dtype` (SGLK_F32 / F16 / BF16).  A scale <= 0
 * means "no scale"; with one, the quotient is formed as the reference's device forms `cache_k.div_(k_scale)`: product
 * with the fp32 reciprocal, rounded to src_dtype, before the fp8 rounding.  Strides in elements of their own arrays.
 * Replaces the fp8 branch of MHATokenToKVPool.set_kv_buffer, memory_pool.py:389-400. */
SGLK_API int sglk_set_kv_buffer_fp8(void* k_buffer, void* v_buffer, const void* k, const void* v,
                                    const void* loc, int loc_is64, int64_t tokens, int row_k, int row_v,
                                    int64_t kbuf_stride_n, int64_t vbuf_stride_n, int64_t k_stride_t,
                                    int64_t v_stride_t, float k_scale, float v_scale, int src_dtype,
                                    int kv_format, void* stream);

/* ------------------------------------------------------------------ E1: elementwise ----------- */

/* Replaces rmsnorm / fused_add_rmsnorm, sgl-kernel/python/sgl_kernel/elementwise.py:9-83
 * (arithmetic: python/sglang/srt/layers/layernorm.py:88-107). */
SGLK_API int sglk_rmsnorm(void* out, const void* x, const void* weight, int64_t rows, int hidden,
                          int64_t x_stride, int64_t out_stride, float eps, int gemma, int dtype,
                          void* stream);
SGLK_API int sglk_fused_add_rmsnorm(void* x, void* residual, const void* weight, int64_t rows,
                                    int hidden, float eps, int gemma, int dtype, void* stream);

/* out[..., :d] = act(x[..., :d]) * x[..., d:]; act: 0 silu, 1 gelu(tanh), 2 gelu(erf).
 * Replaces silu_and_mul & co, sgl-kernel/csrc/elementwise/activation.cu:22-50. */
SGLK_API int sglk_act_and_mul(void* out, const void* x, int64_t rows, int d, int act, int dtype,
                              void* stream);

/* In-place RoPE on q [T,Hq,hs] and k [T,Hk,hs] from cos_sin_cache [max_pos, rot] fp32.
 * Replaces apply_rope_pos_ids_cos_sin_cache, sgl-kernel/csrc/elementwise/rope.cu:22-89. */
SGLK_API int sglk_rope(void* q, void* k, const float* cos_sin_cache, const int64_t* positions,
                       int64_t tokens, int q_heads, int k_heads, int head_size, int rot_dim,
                       int64_t q_stride_t, int64_t k_stride_t, int interleave, int dtype,
                       void* stream);

/* RoPE on q,k of a fused qkv activation [T, (Hq+2Hkv)*hs] and scatter of the rotated k and of v
 * into the pool rows loc[t] in one launch (RadixAttention decode: rotary_emb + set_kv_buffer,
 * python/sglang/srt/models/llama.py:170-175 + memory_pool.py:378-412). */
SGLK_API int sglk_rope_and_store_kv(void* qkv, const float* cos_sin_cache, const int64_t* positions,
                                    void* k_buffer, void* v_buffer, const void* loc, int loc_is64,
                                    int64_t tokens, int q_heads, int kv_heads, int head_size,
                                    int rot_dim, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                                    int interleave, int dtype, void* stream);

/* ------------------------------------------------------------------ G1: AWQ int4 -------------- */

/* out[r, 8c+j] = ((qweight[r,c] >> 4*ORD[j]) & 15) - ((qzeros[r/G,c] >> 4*ORD[j]) & 15)) * scales[r/G,8c+j]
 * ORD = {0,4,1,5,2,6,3,7}.  Replaces awq_dequantize, sgl-kernel/csrc/gemm/awq_kernel.cu:184-219. */
SGLK_API int sglk_awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros,
                                 void* out, int rows, int qcols, int group_size, int dtype,
                                 void* stream);

/* One-time re-layout of an AWQ weight for the fused GEMM (done in process_weights_after_loading).
 * packed: int32 [K/128][N/16][64 lanes][4], the same byte count as qweight; K % 128 == 0, N % 16 == 0. */
SGLK_API int sglk_awq_repack(const int32_t* qweight, int32_t* packed, int rows, int qcols,
                             void* stream);

/* GPTQ int4 checkpoint tensors into the layouts of sglk_awq_repack / the AWQ zero words, so that every sglk_awq_*
 * product runs on them unchanged.  qweight int32 [K/8, N] and qzeros int32 [K/G, N/8] as GPTQLinearMethod registers
 * them (python/sglang/srt/layers/quantization/gptq.py:18-31 takes it from vllm 0.8.4, absent from the reference
 * tree); zero_offset 1 for v1 checkpoints (zero points stored minus one), 0 for v2.  packed int32 [K, N/8],
 * qzeros_out int32 [K/G, N/8]; *overflow (device int, caller-zeroed) becomes non-zero if a zero point does not fit
 * four bits.  Activation reordering (desc_act with a non-trivial g_idx) is not supported. */
SGLK_API int sglk_gptq_repack(const int32_t* qweight, const int32_t* qzeros, int32_t* packed,
                              int32_t* qzeros_out, int* overflow, int K, int N, int group_size,
                              int zero_offset, void* stream);

/* awq_dequantize from the re-laid-out copy: the same values as sglk_awq_dequantize on the original tensor
 * (for the prefill path once the reference-layout qweight has been dropped).  K % 128 == 0, N % 16 == 0. */
SGLK_API int sglk_awq_dequantize_packed(const int32_t* packed, const void* scales, const int32_t* qzeros,
                                        void* out, int K, int N, int group_size, int dtype, void* stream);

/* out[M,N] = x[M,K] @ dequant(W) (+bias) with the int4 weights dequantised in registers and fed to
 * the matrix cores; W in the sglk_awq_repack layout.  Fuses AWQLinearMethod.apply,
 * python/sglang/srt/layers/quantization/awq.py:197-215 (awq_dequantize + torch.matmul).
 * Computes x @ ((q - z) * s) with (q - z) * s never rounded to half precision (fp32 accumulation).
 * K % 128 == 0, N % 64 == 0, group_size a multiple of 128; any M (rows are processed 64 at a time).
 * workspace: sglk_awq_gemm_workspace_bytes() bytes, always required.  Its first 16 KiB are split-K ticket
 * counters: they must be zero before the first call and every call leaves them zero, so one zero-filled
 * buffer serves any number of consecutive calls on a stream (calls that overlap on two streams need two). */
SGLK_API int64_t sglk_awq_gemm_workspace_bytes(int M, int N, int K);
SGLK_API int sglk_awq_gemm(const void* x, const int32_t* packed, const void* scales,
                           const int32_t* qzeros, const void* bias, void* out, float* workspace,
                           int M, int N, int K, int group_size, int64_t x_stride, int64_t out_stride,
                           int dtype, void* stream);

/* sglk_awq_gemm without the workspace argument, for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * split-K workspace is a static device array inside the library (sglk_awq_gemm_static_max_workspace_bytes() =
 * 7 MiB + 16 KiB per GPU the library runs on, held until the process ends: enough for every projection of
 * Llama-3-8B at up to 64 rows); shapes whose
 * sglk_awq_gemm_workspace_bytes() exceeds it are rejected with SGLK_EINVAL.  The call allocates and frees nothing
 * and clears the ticket words itself (a memset node on `stream`).  Every call on one GPU shares that GPU's array:
 * calls that overlap on one GPU (two streams) race -- use sglk_awq_gemm with separate workspaces there.
 * The array belongs to the CURRENT device, which must be the one the operands live on (checked: SGLK_EINVAL
 * otherwise).  sglk_awq_gemm_static_init() does, for the current device, everything a first call would do (the array's
 * address, the kernels' LDS limits); it is REQUIRED before a call whose stream is being captured -- a first use on a
 * device under capture is refused with SGLK_EINVAL rather than resolved lazily. */
SGLK_API int64_t sglk_awq_gemm_static_max_workspace_bytes(void);
SGLK_API int sglk_awq_gemm_static_init(void);
SGLK_API int sglk_awq_gemm_static(const void* x, const int32_t* packed, const void* scales,
                                  const int32_t* qzeros, const void* bias, void* out, int M, int N, int K,
                                  int group_size, int64_t x_stride, int64_t out_stride, int dtype,
                                  void* stream);

/* ---- fusion forms of the decode layer --------------------------------------------------------------
 * None of these has a counterpart in the reference, which runs norm / GEMM / rope / activation as separate
 * ops; they are for callers that own the whole decoder layer (the graph-captured decode step).  Two ideas:
 *  (1) block sums: the fused AWQ GEMM needs X[g][m] = sum of x[m, 128g .. 128g+127]; the kernel that produces
 *      x can write them (xsum [K/128][xs_stride] fp32, xs_stride >= M) and the GEMM then skips its own pass;
 *  (2) split-K partials: the GEMM stores fp32 partial[split][M][N] (row-major, split stride M*N) and the
 *      consumer adds the splits up, rounds once to the activation type (the value the GEMM would have
 *      stored) and carries on -- no last-block reduction inside the GEMM. */
/* sglk_awq_gemm with the block sums of x supplied (xsum may be NULL: computed in the kernel). */
SGLK_API int sglk_awq_gemm_xs(const void* x, const int32_t* packed, const void* scales,
                              const int32_t* qzeros, const void* bias, void* out, float* workspace, int M,
                              int N, int K, int group_size, int64_t x_stride, int64_t out_stride, int dtype,
                              const float* xsum, int64_t xs_stride, void* stream);
/* Number of splits sglk_awq_gemm_partial writes for this shape; 0: shape not covered by the partial form
 * (needs K % 512 == 0, N % 128 == 0, group size 128). */
SGLK_API int sglk_awq_gemm_partial_splits(int M, int N, int K, int group_size);
SGLK_API int sglk_awq_gemm_partial(const void* x, const int32_t* packed, const void* scales,
                                   const int32_t* qzeros, const float* xsum, int64_t xs_stride,
                                   float* partial, int M, int N, int K, int group_size, int64_t x_stride,
                                   int dtype, void* stream);
/* fused_add_rmsnorm (residual may be NULL: plain rmsnorm in place) whose input row may be given as split-K
 * partials (+bias) instead of x, and which may write the block sums of its output.  16-bit types,
 * hidden % 128 == 0, hidden <= 8192.  zero_out / zero_count: fp32 words cleared on the way (the accumulation
 * buffer of a later kernel), or NULL / 0. */
SGLK_API int sglk_fused_add_rmsnorm_ex(void* x, void* residual, const void* weight, int64_t rows, int hidden,
                                       float eps, int gemma, int dtype, const float* partials, int splits,
                                       int64_t part_stride, const void* bias, float* xsum_out,
                                       int64_t xs_stride, float* zero_out, int64_t zero_count,
                                       void* stream);
/* gate_up projection with the gated activation in its epilogue: out[M, N/2] = silu(x @ W_gate) * (x @ W_up),
 * rounded exactly where sglk_awq_gemm followed by sglk_act_and_mul rounds.  W must have been re-laid-out from a
 * weight whose columns were interleaved in blocks of 64: new columns [128t, 128t+64) = gate columns
 * [64t, 64t+64), new columns [128t+64, 128t+128) = up columns [64t, 64t+64) (scales and zeros likewise).
 * xsum (may be NULL): block sums of x.  xsum_out (may be NULL): block sums of the output rows are ADDED to it
 * (it must be zero on entry: sglk_fused_add_rmsnorm_ex can clear it with zero_out / zero_count).
 * N % 256 == 0, K % 512 == 0, group size 128. */
SGLK_API int sglk_awq_gemm_silu_mul(const void* x, const int32_t* packed, const void* scales,
                                    const int32_t* qzeros, void* out, int M, int N, int K, int group_size,
                                    int64_t x_stride, int64_t out_stride, int dtype, const float* xsum,
                                    int64_t xs_stride, float* xsum_out, int64_t xs_out_stride,
                                    void* stream);
/* Several projections of one decoder layer in ONE launch: job i + 1 multiplies what job i wrote.  The layer's
 * projections (layers/quantization/awq.py:197-215 per linear; models/llama.py:245-267 for their order) each cost a
 * launch with its own fill (the first weights and activations on their way, 3-4 us) and drain; here one persistent
 * block per CU walks the jobs, requests the next job's first weights, scales and zeros before it finishes the
 * current one, and hands over through write-through stores and relaxed counters (no fence, no kernel boundary).
 * kind SGLK_CHAIN_PARTIAL: what sglk_awq_gemm_partial does (fp32 split sums to `partial`, splits as
 * sglk_awq_gemm_partial_splits says; splits_expected, when non-zero, must equal that count).
 * kind SGLK_CHAIN_SILU_MUL: what sglk_awq_gemm_silu_mul does (out, out_stride, xsum_out, xs_out_stride).
 * Every job takes the block sums of its x from its producer (xsum_in, xs_stride >= M).  M <= 64; at most 8 jobs;
 * shapes as the two entry points require; every job's column tiles x splits must fit the device's CUs.
 * Same bits as the separate launches.  A job may read what an EARLIER job of the chain wrote only if no job before
 * that writer read the same addresses (the L2s of the XCDs are not coherent inside a launch): distinct buffers per
 * hand-over.  sync: sglk_awq_chain_sync_bytes() bytes of device memory, ZERO before the first launch that uses it
 * (every launch leaves it zero); one chain at a time per sync buffer.  A block that waits two seconds for its
 * producers raises the error word (sglk_awq_chain_status, synchronising: 1) and goes on rather than hang. */
#define SGLK_CHAIN_PARTIAL 0
#define SGLK_CHAIN_SILU_MUL 1
typedef struct sglk_chain_job {
  int kind;
  int N, K;
  int splits_expected;
  const void* x;
  int64_t x_stride;
  const int32_t* packed;
  const void* scales;
  const int32_t* qzeros;
  const float* xsum_in;
  int64_t xs_stride;
  void* out;
  int64_t out_stride;
  float* partial;
  float* xsum_out;
  int64_t xs_out_stride;
} sglk_chain_job;
SGLK_API int64_t sglk_awq_chain_sync_bytes(void);
SGLK_API int sglk_awq_chain_status(const void* sync);
SGLK_API int sglk_awq_chain(const sglk_chain_job* jobs, int njobs, void* sync, int M, int group_size, int dtype,
                            void* stream);
/* sglk_act_and_mul that also writes the block sums of its output rows (d % 128 == 0). */
SGLK_API int sglk_act_and_mul_xs(void* out, const void* x, int64_t rows, int d, int act, int dtype,
                                 float* xsum_out, int64_t xs_stride, void* stream);
/* sglk_rope_and_store_kv on a qkv row first assembled from split-K partials (written to qkv). */
SGLK_API int sglk_rope_and_store_kv_partial(void* qkv, const float* partials, int splits,
                                            int64_t part_stride, const void* bias,
                                            const float* cos_sin_cache, const int64_t* positions,
                                            void* k_buffer, void* v_buffer, const void* loc, int loc_is64,
                                            int64_t tokens, int q_heads, int kv_heads, int head_size,
                                            int rot_dim, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                                            int interleave, int dtype, void* stream);
/* Decode attention that also writes the block sums of its output rows (Dv == 128: one head per block;
 * xsum_out [Hq][xs_stride], xs_stride >= B). */
SGLK_API int sglk_decode_attention_xs(const void* q, const void* k_buffer, const void* v_buffer, void* o,
                                      const int32_t* kv_indptr, const int32_t* kv_indices,
                                      float* attn_logits, float* attn_lse, const int32_t* num_kv_splits,
                                      int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv,
                                      int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_n,
                                      int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
                                      int64_t o_stride_b, int64_t o_stride_h, float sm_scale,
                                      float logit_cap, int dtype, float* xsum_out, int64_t xs_stride,
                                      void* stream);
SGLK_API int sglk_decode_attention_shared_prefix_xs(
    const void* q, const void* k_buffer, const void* v_buffer, void* o, const int32_t* kv_indptr,
    const int32_t* kv_indices, const int32_t* group_indptr, const int32_t* prefix_lens, int G,
    float* workspace, int max_kv_splits, int B, int Hq, int Hkv, int D, int Dv, int64_t q_stride_b,
    int64_t q_stride_h, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n, int64_t v_stride_h,
    int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype, float* xsum_out,
    int64_t xs_stride, void* stream);
/* sglk_rope_and_store_kv_partial folded into sglk_decode_attention_shared_prefix_xs (the reference runs
 * rotary_embedding, set_kv_buffer and decode_attention_fwd as three ops: models/llama.py:160-190,
 * layers/radix_attention.py:60-86): the query rows and every request's new K / V row are assembled from the
 * qkv projection's split-K partial sums inside the attention kernel -- summed in split order, biased, rounded,
 * rotated, rounded: the bits the separate kernel stores -- the new rows are written to pool slot loc[b], which
 * must be the last entry of request b's index list, and no rotated q is written anywhere.
 * Rotation of full 128-element heads in the (d, d + 64) pairing only (rot_dim == 128, interleave == 0), at most
 * 16 query heads per kv head; SGLK_EUNSUP otherwise. */
SGLK_API int sglk_decode_attention_shared_prefix_rope(
    const float* qkv_partials, int splits, int64_t part_stride, const void* qkv_bias,
    const float* cos_sin_cache, const int64_t* positions, int rot_dim, int interleave, const void* loc,
    int loc_is64, void* k_buffer, void* v_buffer, void* o, const int32_t* kv_indptr, const int32_t* kv_indices,
    const int32_t* group_indptr, const int32_t* prefix_lens, int G, float* workspace, int max_kv_splits, int B,
    int Hq, int Hkv, int D, int Dv, int64_t k_stride_n, int64_t k_stride_h, int64_t v_stride_n,
    int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, float logit_cap, int dtype,
    float* xsum_out, int64_t xs_stride, void* stream);

/* Skinny dense GEMM out[M,N] = x[M,K] @ W[N,K]^T (+bias), bf16/fp16, for the unquantised lm_head and
 * the bf16 configuration (UnquantizedLinearMethod.apply, python/sglang/srt/layers/linear.py:170-186;
 * CPU twin weight_packed_linear, sgl-kernel/csrc/cpu/gemm.cpp). */
SGLK_API int sglk_dense_gemm_nt(const void* x, const void* w, const void* bias, void* out, int M, int N,
                                int K, int64_t x_stride, int64_t w_stride, int64_t out_stride,
                                int dtype, void* stream);
/* Large-M (extend / prefill) products, out[M, N] = x[M, K] @ W (+ bias) on 256 x 128 output tiles with
 * v_mfma_f32_32x32x16 and two LDS stages.  sglk_awq_gemm_prefill dequantises the packed int4 weight on the way into
 * LDS -- (q - z) * scale rounded to the activation type, i.e. exactly the matrix the reference's
 * awq_dequantize-then-matmul multiplies by (python/sglang/srt/layers/quantization/awq.py:197-215) -- so no
 * dequantised copy is written to memory; sglk_dense_gemm_prefill takes a 16-bit weight [N, K].
 * K % 128 == 0, N % 128 == 0, group_size % 128 == 0. */
SGLK_API int sglk_awq_gemm_prefill(const void* x, const int32_t* packed, const void* scales, const int32_t* qzeros,
                                   const void* bias, void* out, int M, int N, int K, int group_size,
                                   int64_t x_stride, int64_t out_stride, int dtype, void* stream);
SGLK_API int sglk_dense_gemm_prefill(const void* x, const void* w, const void* bias, void* out, int M, int N, int K,
                                     int64_t x_stride, int64_t w_stride, int64_t out_stride, int dtype,
                                     void* stream);
/* The same product on a weight re-laid-out once at load time for sequential streaming (every wave instruction
 * reads 1 KiB of consecutive memory): packed = [N / 16][K / 128][4][64 lanes][16 bytes], same byte count as w.
 * For weights that are read whole at every step and never through any other kernel (the lm_head:
 * python/sglang/srt/layers/vocab_parallel_embedding.py ParallelLMHead -> logits_processor.py _get_logits).
 * N % 16 == 0, K % 128 == 0.  No counterpart in the reference, which hands the row-major weight to a library GEMM. */
SGLK_API int sglk_dense_repack(const void* w, void* packed, int N, int K, int64_t w_stride, void* stream);
SGLK_API int sglk_dense_gemm_packed(const void* x, const void* packed, const void* bias, void* out, int M, int N,
                                    int K, int64_t x_stride, int64_t out_stride, int dtype, void* stream);
/* Split-K forms of the packed product for weights of a few thousand rows (the decoder-layer projections of a 16-bit
 * model, layers/linear.py UnquantizedLinearMethod.apply: one block per 64 columns would leave most CUs idle).
 * _ws: workspace of sglk_dense_gemm_workspace_bytes() bytes (contents don't matter), fp32 slabs summed in split order
 * by a second launch; falls back to the unsplit form when the shape does not split or the workspace is too small.
 * _partial: the slabs partial[split][M][N] (sglk_dense_gemm_partial_splits(M, N, K) of them, 0 = shape not covered) are
 * left to the consumer kernel (sglk_fused_add_rmsnorm_ex, sglk_rope_and_store_kv_partial), which adds them up in split
 * order and rounds once: the value the GEMM would have stored.  No counterpart in the reference. */
SGLK_API int sglk_dense_gemm_partial_splits(int M, int N, int K);
SGLK_API int sglk_dense_gemm_packed_partial(const void* x, const void* packed, float* partial, int M, int N, int K,
                                            int64_t x_stride, int dtype, void* stream);
SGLK_API int sglk_dense_gemm_packed_ws(const void* x, const void* packed, const void* bias, void* out, int M, int N,
                                       int K, int64_t x_stride, int64_t out_stride, void* workspace,
                                       int64_t workspace_bytes, int dtype, void* stream);
/* gate_up product with the gated activation in the launch that sums the splits: packed = the [2d, K] weight (gate
 * rows, then up rows), out [M, d] = round(round(act(round(gate))) * round(up)), the values the product followed by
 * sglk_act_and_mul stores (layers/activation.py SiluAndMul / GeluAndMul over the merged projection).  act: 0 silu,
 * 1 gelu (tanh), 2 gelu.  workspace >= sglk_dense_gemm_partial_splits(M, N, K) * M * N * 4 bytes. */
SGLK_API int sglk_dense_gemm_packed_act(const void* x, const void* packed, void* out, int M, int N, int K,
                                        int64_t x_stride, int64_t out_stride, int act, void* workspace,
                                        int64_t workspace_bytes, int dtype, void* stream);
/* The same product with a scratch buffer (sglk_dense_gemm_workspace_bytes(), contents don't matter): narrow
 * outputs (the decoder-layer projections of the bf16 configuration) are split along K over enough blocks to fill
 * the chip, fp32 partials summed in split order by a second launch.  Falls back to the form above when the
 * workspace is missing or too small. */
SGLK_API int64_t sglk_dense_gemm_workspace_bytes(int M, int N, int K);
SGLK_API int sglk_dense_gemm_nt_ws(const void* x, const void* w, const void* bias, void* out, int M, int N,
                                   int K, int64_t x_stride, int64_t w_stride, int64_t out_stride,
                                   void* workspace, int64_t workspace_bytes, int dtype, void* stream);

/* ------------------------------------------------------------------ G2/G3: fp8 ---------------- */

/* Replaces sgl_per_token_quant_fp8, sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:78-106. */
SGLK_API int sglk_per_token_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int hidden,
                                      int dtype, void* stream);
/* Replaces sgl_per_token_group_quant_fp8, sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:96-197.
 * scale_col_major: scales laid out [hidden/group, rows] instead of [rows, hidden/group]. */
SGLK_API int sglk_per_token_group_quant_fp8(const void* x, void* q, float* scales, int64_t rows,
                                            int hidden, int group_size, float eps, float fp8_min,
                                            float fp8_max, int scale_col_major, int dtype,
                                            void* stream);
/* Replaces sgl_per_tensor_quant_fp8, sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu. */
SGLK_API int sglk_per_tensor_quant_fp8(const void* x, void* q, float* scale, int64_t numel,
                                       int is_static, int dtype, void* stream);
/* The same quantisation of a [rows, cols] view whose rows lie x_row_stride elements apart (q: q_row_stride bytes
 * apart), one scale for the whole view: what the reference reaches with per_tensor_quant_mla_fp8 on the sliced,
 * transposed q_nope (python/sglang/srt/models/deepseek_v2.py:965-972) without the copy in between.  The dynamic
 * form takes a running maximum into `scale`, which must hold 0 on entry. */
SGLK_API int sglk_per_tensor_quant_fp8_rows(const void* x, void* q, float* scale, int64_t rows, int64_t cols,
                                            int64_t x_row_stride, int64_t q_row_stride, int is_static,
                                            int dtype, void* stream);
/* out = (A @ B) * scales_a[M] * scales_b[N] (+bias); A [M,K] e4m3 row-major, B given as [N,K]
 * row-major (the reference's column-major [K,N]).  Replaces fp8_scaled_mm,
 * sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1071-1146. */
SGLK_API int sglk_fp8_scaled_mm(const void* a, const void* b_nk, const float* scales_a,
                                const float* scales_b, const void* bias, void* out, int M, int N,
                                int K, int64_t a_stride, int64_t b_stride, int64_t out_stride,
                                int out_dtype, void* stream);
/* Producers of a w8a8 linear's input that quantise on the way out (for callers that own the whole decoder layer):
 * fused_add_rmsnorm (residual += x when residual is given; sgl-kernel/csrc/elementwise: norm via
 * python/sgl_kernel/elementwise.py:10-75) resp. silu_and_mul (elementwise.py:78-100, x = [gate | up] of width 2d)
 * followed by sgl_per_token_quant_fp8 (gemm.py:131-136), with the roundings of that sequence; the 16-bit tensor
 * in between is never written.  16-bit x, q e4m3 [rows, hidden] resp. [rows, d], scales fp32 [rows]. */
SGLK_API int sglk_rmsnorm_quant_fp8(const void* x, void* residual, const void* weight, void* q,
                                    float* scales, int64_t rows, int hidden, float eps, int gemma,
                                    int dtype, void* stream);
SGLK_API int sglk_silu_and_mul_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int d,
                                         int dtype, void* stream);
/* The same two fusions with per (token, 128-element group) scales (sgl_per_token_group_quant_fp8,
 * sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:96-197, group 128, eps 1e-10, +-448): what the block-fp8 linear
 * method quantises its input with (python/sglang/srt/layers/quantization/fp8_utils.py:190-197) folded into the norm /
 * activation in front of it.  scales [rows, hidden / 128] row-major; x rows x_row_stride elements apart. */
SGLK_API int sglk_rmsnorm_group_quant_fp8(const void* x, void* residual, const void* weight, void* q, float* scales,
                                          int64_t rows, int hidden, int64_t x_row_stride, float eps, int gemma,
                                          int dtype, void* stream);
SGLK_API int sglk_silu_and_mul_group_quant_fp8(const void* x, void* q, float* scales, int64_t rows, int d,
                                               int dtype, void* stream);
/* The same product with a scratch buffer (sglk_fp8_mm_workspace_bytes(), contents don't matter): narrow outputs
 * (N = 1280 of a TP = 8 qkv shard gives 40 blocks) are split along K, fp32 partials summed in split order and scaled
 * by a second launch.  Falls back to the form above when the workspace is missing or too small. */
SGLK_API int64_t sglk_fp8_mm_workspace_bytes(int M, int N, int K);
SGLK_API int sglk_fp8_scaled_mm_ws(const void* a, const void* b_nk, const float* scales_a,
                                   const float* scales_b, const void* bias, void* out, int M, int N,
                                   int K, int64_t a_stride, int64_t b_stride, int64_t out_stride,
                                   void* workspace, int64_t workspace_bytes, int out_dtype,
                                   void* stream);
/* The same product on a weight re-laid-out once at load time for sequential streaming (the decode step reads every
 * weight whole, once per step): packed = [N / 16][K / 128][2][64 lanes][16 bytes], same byte count as b_nk, made by
 * sglk_fp8_repack (N % 16 == 0, K % 128 == 0).  A wave streams its strip of 16 weight rows straight into the
 * matrix-core operands, the quantised activations go through LDS; narrow outputs are split along K through the
 * workspace (sglk_fp8_mm_packed_workspace_bytes(), contents don't matter; without it the product runs unsplit).
 * Same arithmetic contract as sglk_fp8_scaled_mm (fp32 accumulation, scales_a[m] * scales_b[n], + bias).  No
 * counterpart in the reference, which hands the row-major weight to cutlass / torch._scaled_mm
 * (layers/quantization/fp8_utils.py:454-693). */
SGLK_API int sglk_fp8_repack(const void* w_nk, void* packed, int N, int K, int64_t w_stride, void* stream);
SGLK_API int64_t sglk_fp8_mm_packed_workspace_bytes(int M, int N, int K);
SGLK_API int sglk_fp8_scaled_mm_packed(const void* a, const void* packed, const float* scales_a,
                                       const float* scales_b, const void* bias, void* out, int M, int N,
                                       int K, int64_t a_stride, int64_t out_stride, void* workspace,
                                       int64_t workspace_bytes, int out_dtype, void* stream);
/* The block-scaled product (sglk_fp8_blockwise_scaled_mm: a contiguous [M, K], out contiguous [M, N]) on a weight from
 * sglk_fp8_repack; scale arguments and strides as there.  Narrow outputs split along K through the workspace
 * (sglk_fp8_mm_packed_workspace_bytes()), without one the product runs unsplit. */
SGLK_API int sglk_fp8_blockwise_scaled_mm_packed(const void* a, const void* packed, const float* scales_a,
                                                 const float* scales_b, void* out, int M, int N, int K,
                                                 int64_t sa_stride_m, int64_t sa_stride_k, int64_t sb_stride_n,
                                                 int64_t sb_stride_k, void* workspace, int64_t workspace_bytes,
                                                 int out_dtype, void* stream);
/* The streaming product's raw split sums partial[split][M][N] (sglk_fp8_mm_packed_partial_splits(M, N, K) slabs, 0 =
 * shape not covered; one slab when the shape does not split) left to the kernel that consumes the product:
 * sglk_rmsnorm_quant_fp8_partial, sglk_silu_and_mul_quant_fp8_partial (rows of 2 d columns: gate, then up; dtype = the
 * 16-bit type the product would have stored) and sglk_rope_and_store_kv_partial_scaled form
 * round(sum * scales_a[m] * scales_b[n] + bias[n]) exactly as the reduction launch of sglk_fp8_scaled_mm_packed does
 * and carry on (fused add + norm + per-token quantisation; rotation + pool write).  row_scale may alias the `scales`
 * output of sglk_rmsnorm_quant_fp8_partial.  No counterpart in the reference, which runs these steps as separate ops. */
SGLK_API int sglk_fp8_mm_packed_partial_splits(int M, int N, int K);
SGLK_API int sglk_fp8_mm_packed_partial(const void* a, const void* packed, float* partial, int M, int N, int K,
                                        int64_t a_stride, void* stream);
SGLK_API int sglk_rmsnorm_quant_fp8_partial(const float* partials, int splits, int64_t part_stride,
                                            const float* row_scale, const float* col_scale, const void* bias,
                                            void* residual, const void* weight, void* q, float* scales,
                                            int64_t rows, int hidden, float eps, int gemma, int dtype, void* stream);
SGLK_API int sglk_silu_and_mul_quant_fp8_partial(const float* partials, int splits, int64_t part_stride,
                                                 const float* row_scale, const float* col_scale, void* q,
                                                 float* scales, int64_t rows, int d, int dtype, void* stream);
SGLK_API int sglk_rope_and_store_kv_partial_scaled(void* qkv, const float* partials, int splits, int64_t part_stride,
                                                   const void* bias, const float* row_scale, const float* col_scale,
                                                   const float* cos_sin_cache, const int64_t* positions,
                                                   void* k_buffer, void* v_buffer, const void* loc, int loc_is64,
                                                   int64_t tokens, int q_heads, int kv_heads, int head_size,
                                                   int rot_dim, int64_t kbuf_stride_n, int64_t vbuf_stride_n,
                                                   int interleave, int dtype, void* stream);
/* DeepSeek block-scaled form: scales_a [M, K/128] (stride given, row- or M-major), scales_b
 * [N/128, K/128] or its transpose.  Replaces fp8_blockwise_scaled_mm,
 * sgl-kernel/csrc/gemm/fp8_blockwise_gemm_kernel.cu:345-423. */
SGLK_API int sglk_fp8_blockwise_scaled_mm(const void* a, const void* b_nk, const float* scales_a,
                                          const float* scales_b, void* out, int M, int N, int K,
                                          int64_t sa_stride_m, int64_t sa_stride_k,
                                          int64_t sb_stride_n, int64_t sb_stride_k, int out_dtype,
                                          void* stream);

/* Batched fp8 matmul with one scale per operand: out[b] = (A[b] @ B[b]) * scale_a[0] * scale_b[0], A [batch, M, K]
 * row-major, B given as [batch, N, K] row-major (the reference's column-major [batch, K, N]); either operand (not
 * both) may be e5m2.  Replaces bmm_fp8, sgl-kernel/csrc/gemm/bmm_fp8.cu:23-76 (python/sgl_kernel/gemm.py:66-82),
 * which DeepSeek's absorbed MLA uses for the per-head q / o projections. */
SGLK_API int sglk_bmm_fp8(const void* a, const void* b_nk, const float* scale_a, const float* scale_b,
                          void* out, int batch, int M, int N, int K, int64_t a_batch_stride,
                          int64_t a_stride, int64_t b_batch_stride, int64_t b_stride,
                          int64_t out_batch_stride, int64_t out_stride, int a_e5m2, int b_e5m2,
                          int out_dtype, void* stream);

/* ------------------------------------------------------------------ MoE routing helpers ------- */

/* Group the numel (token, k) slots of topk_ids by expert, every expert's run padded to a multiple of
 * block_size: sorted_token_ids (pre-filled by the caller with numel, the padding value), expert_ids
 * [num_tokens_post_pad / block_size], num_tokens_post_pad [1]; cumsum_buffer [num_experts + 1] (optional)
 * receives the padded prefix sums.  Slots of one expert keep their order of appearance.  Replaces
 * moe_align_block_size, sgl-kernel/csrc/moe/moe_align_kernel.cu:160-227 (torch_extension_rocm.cc:60-63). */
SGLK_API int sglk_moe_align_block_size(const void* topk_ids, int ids_is64, int64_t numel,
                                       int num_experts, int block_size, int32_t* sorted_token_ids,
                                       int32_t* expert_ids, int32_t* num_tokens_post_pad,
                                       int32_t* cumsum_buffer, void* stream);
/* Softmax over the experts of gating_output [num_tokens, num_experts], then the topk largest per row:
 * topk_weights fp32, topk_ids int32, token_expert_indices[t, j] = j * num_tokens + t.  Replaces topk_softmax,
 * sgl-kernel/csrc/moe/moe_topk_softmax_kernels.cu:477-505 (torch_extension_rocm.cc:65-68). */
SGLK_API int sglk_topk_softmax(const void* gating_output, float* topk_weights, int32_t* topk_ids,
                               int32_t* token_expert_indices, int num_tokens, int num_experts,
                               int topk, int64_t gating_stride, int dtype, void* stream);

/* Grouped, bias-corrected top-k gate: moe_fused_gate, sgl-kernel/csrc/moe/moe_fused_gate.cu:52-245,362-500
 * (== biased_grouped_topk, python/sglang/srt/layers/moe/topk.py:155-215).  input [rows, num_experts] and bias
 * [num_experts] in fp32 / fp16 / bf16; output fp32 and indices int32 [rows, topk].  n_share_experts_fusion > 0: the
 * last slot is the shared expert num_experts + row % n_share with weight sum / routed_scaling_factor.  Any
 * num_experts up to 1024 that num_expert_group (<= 64) divides (the reference: powers of two, 32 per group at most). */
SGLK_API int sglk_moe_fused_gate(const void* input, const void* bias, float* output, int32_t* indices,
                                 int num_rows, int num_experts, int num_expert_group, int topk_group,
                                 int topk, int n_share_experts_fusion, float routed_scaling_factor,
                                 int64_t input_stride, int dtype, void* stream);

/* The block-scaled product with the same scratch buffer (sglk_fp8_mm_workspace_bytes()): K split over blocks for
 * narrow outputs; the partials are already scaled, the second launch adds them up in split order. */
SGLK_API int sglk_fp8_blockwise_scaled_mm_ws(const void* a, const void* b_nk, const float* scales_a,
                                             const float* scales_b, void* out, int M, int N, int K,
                                             int64_t sa_stride_m, int64_t sa_stride_k,
                                             int64_t sb_stride_n, int64_t sb_stride_k, void* workspace,
                                             int64_t workspace_bytes, int out_dtype, void* stream);

/* ------------------------------------------------------------------ sampling filters ---------- */

/* probs fp32 [batch, vocab] (rows sum to 1).  Keep the top_k largest entries of every row (ties with the k-th kept)
 * and renormalise: top_k_renorm_probs, sgl-kernel/python/sgl_kernel/sampling.py:21-52.  top_k_arr (int32 [batch]) or,
 * when null, top_k_val for every row. */
SGLK_API int sglk_top_k_renorm_probs(const float* probs, float* out, const int32_t* top_k_arr,
                                     int top_k_val, int batch, int vocab, int64_t in_stride,
                                     int64_t out_stride, void* stream);
/* Drop the longest ascending run of entries whose cumulative mass stays below 1 - top_p, renormalise the rest:
 * top_p_renorm_probs, sampling.py:69-102.  The cumulative masses are fp32 sums whose order is not fixed: an entry
 * whose cumulative mass sits within rounding of 1 - top_p may fall on either side of the cut from run to run. */
SGLK_API int sglk_top_p_renorm_probs(const float* probs, float* out, const float* top_p_arr,
                                     float top_p_val, int batch, int vocab, int64_t in_stride,
                                     int64_t out_stride, void* stream);
/* One token per row from the filtered distribution by inverse CDF in index order with the caller's uniform numbers
 * (fp32 [batch] in [0, 1)): top_p_sampling_from_probs / top_k_top_p_sampling_from_probs / min_p_sampling_from_probs,
 * sampling.py:131-330.  A filter is off when its value is out of range (top_k <= 0 or >= vocab, top_p >= 1,
 * min_p <= 0); joint != 0 applies top-k and top-p to the original distribution ("joint"), 0 applies top-p to the
 * top-k-renormalised one ("top_k_first").  out int32 [batch]. */
SGLK_API int sglk_sampling_from_probs(const float* probs, const float* uniform, int32_t* out,
                                      const int32_t* top_k_arr, int top_k_val,
                                      const float* top_p_arr, float top_p_val,
                                      const float* min_p_arr, float min_p_val, int joint, int batch,
                                      int vocab, int64_t in_stride, void* stream);

/* ------------------------------------------------------------------ draft-tree helpers -------- */

/* Layout of the draft trees of a batch: build_tree_kernel_efficient,
 * sgl-kernel/csrc/speculative/eagle_utils.cu:32-147.  parent_list int64 [bs, topk * (depth - 1) + 1],
 * selected_index int64 [bs, n - 1] (n = draft_token_num), verified_seq_len int32 [bs]; outputs: tree_mask bool
 * (1 byte) [sum(seq_len) * n + bs * n * n] (caller-filled with true; the tree part of every row is rewritten),
 * positions int64 [bs * n], retrive_index / retrive_next_token / retrive_next_sibling int64 [bs, n] (caller-filled
 * with -1; entries without a child / sibling are left untouched).  A token whose parent is not among the selected
 * ones (the reference prints a warning and skips it) gets no links and masks only itself.  tree_mask_len = entries
 * of tree_mask; a request whose rows would not fit writes nothing. */
SGLK_API int sglk_build_tree_efficient(const int64_t* parent_list, const int64_t* selected_index,
                                       const int32_t* verified_seq_len, uint8_t* tree_mask,
                                       int64_t* positions, int64_t* retrive_index,
                                       int64_t* retrive_next_token, int64_t* retrive_next_sibling,
                                       int topk, int depth, int draft_token_num, int bs,
                                       int64_t tree_mask_len, void* stream);
/* Greedy verification of the trees against the target model's picks: verify_tree_greedy, eagle_utils.cu:150-286.
 * All int32; predicts [total draft tokens], accept_index [bs, num_spec_step], accept_token_num [bs],
 * candidates / retrive_* / target_predict [bs, num_draft_tokens] (target_predict addressed by flat retrive index);
 * total_draft_tokens = entries of predicts (= of target_predict).  Link or index values outside their range end the
 * walk of that request (the reference reads through them). */
SGLK_API int sglk_verify_tree_greedy(int32_t* predicts, int32_t* accept_index, int32_t* accept_token_num,
                                     const int32_t* candidates, const int32_t* retrive_index,
                                     const int32_t* retrive_next_token,
                                     const int32_t* retrive_next_sibling, const int32_t* target_predict,
                                     int bs, int num_spec_step, int num_draft_tokens,
                                     int total_draft_tokens, void* stream);
/* Target-only speculative sampling over the trees: tree_speculative_sampling_target_only,
 * sgl-kernel/csrc/speculative/speculative_sampling.cuh:38-157 (threshold_acc is floored at 1e-9 as at :185).
 * uniform_samples fp32 [bs, num_draft_tokens], target_probs / draft_probs fp32 [bs, num_draft_tokens, vocab];
 * draft_probs receives the target probability of every rejected candidate. */
SGLK_API int sglk_tree_speculative_sampling_target_only(
    int32_t* predicts, int32_t* accept_index, int32_t* accept_token_num, const int32_t* candidates,
    const int32_t* retrive_index, const int32_t* retrive_next_token, const int32_t* retrive_next_sibling,
    const float* uniform_samples, const float* target_probs, float* draft_probs, int bs, int num_spec_step,
    int num_draft_tokens, int total_draft_tokens, int vocab, float threshold_single, float threshold_acc,
    void* stream);

/* Per-segment bit packing, least significant bit first: segment_packbits, sgl-kernel/csrc/speculative/packbit.cu:25-48
 * (flashinfer's SegmentPackBits with BitOrder::kLittle; the draft-tree mask for backends that take it packed).
 * x bool (1 byte) flat, input_indptr / output_indptr int32 [batch + 1], y uint8; segment i writes
 * ceil(len_i / 8) bytes at output_indptr[i] (never past output_indptr[i + 1]). */
SGLK_API int sglk_segment_packbits(const uint8_t* x, const int32_t* input_indptr,
                                   const int32_t* output_indptr, uint8_t* y, int batch, void* stream);

/* ------------------------------------------------------------------ C1: custom all-reduce ----- */

/* One-/two-stage peer-to-peer all-reduce over xGMI on IPC-mapped buffers.
 * Replaces the ROCm op set of sgl-kernel/csrc/torch_extension_rocm.cc:25-55
 * (python/sgl_kernel/allreduce.py:5-50; device code custom_all_reduce_hip.cuh:155-575).
 * The set-up calls below (init, register, IPC, allocation) are host-side and may allocate or
 * synchronise; sglk_ar_all_reduce itself only launches. */
SGLK_API int64_t sglk_ar_meta_size(void);
/* meta_ptrs[w]: every rank's signal block as mapped into THIS process: zero-initialised uncached
 * device memory of sglk_ar_meta_size() + staging_bytes bytes (the staging area of the two-stage
 * algorithm follows the signals).  Returns an opaque handle (0 on failure).  At most 8 ranks.
 * (init_custom_ar, sgl-kernel/csrc/allreduce/custom_all_reduce.hip:13-49) */
SGLK_API int64_t sglk_ar_init(void* const* meta_ptrs, int rank, int world, int full_xgmi,
                              int64_t staging_bytes);
/* The same with the peers' signal blocks named by IPC handles, as the reference's ROCm binding takes them
 * (init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink), custom_all_reduce.hip:13-35;
 * constructor custom_all_reduce_hip.cuh:376-399): handles[w] points at the sglk_ipc_handle_bytes() bytes of rank
 * w's handle, offsets[w] is that rank's block offset inside the allocation; the own entries are not opened.
 * meta is this rank's block (sglk_ar_meta_size() + staging_bytes bytes of zeroed uncached memory).  rank_data /
 * rank_data_bytes: caller-owned device memory for the records of peer pointers (64 bytes per registered or
 * captured input; the reference hands in 8 MiB, custom_all_reduce.py:289-291); NULL makes the communicator
 * allocate room for 4096 records itself.  Mappings opened here are closed by sglk_ar_dispose. */
SGLK_API int64_t sglk_ar_init_handles(void* meta, void* rank_data, int64_t rank_data_bytes,
                                      const void* const* handles, const int64_t* offsets, int rank, int world,
                                      int full_xgmi, int64_t staging_bytes);
SGLK_API int sglk_ar_dispose(int64_t fa);
/* data_ptrs[w]: every rank's registered input buffer for `local` (peer mappings);
 * data_ptrs[rank] == local.  (register_buffer, custom_all_reduce.hip:119-144) */
SGLK_API int sglk_ar_register_buffer(int64_t fa, void* local, void* const* data_ptrs);
/* The same with the peers' buffers named by IPC handle + offset
 * (register_buffer(fa, t, handles, offsets), custom_all_reduce.hip:119-124, custom_all_reduce_hip.cuh:442-458). */
SGLK_API int sglk_ar_register_buffer_handles(int64_t fa, void* local, const void* const* handles,
                                             const int64_t* offsets);
/* Graph capture (custom_all_reduce_hip.cuh:413-434,460-476; caller custom_all_reduce.py:335-348,388-393): while
 * the stream is capturing, sglk_ar_all_reduce accepts an input that is not registered, reserves a record for it and
 * remembers the pointer.  After the capture: sglk_ar_graph_buffer_count() inputs are pending;
 * sglk_ar_get_graph_buffer_ipc_meta writes their IPC handles (count * sglk_ipc_handle_bytes() bytes, in launch
 * order) and offsets (get_graph_buffer_ipc_meta, custom_all_reduce.hip:126-137); the caller exchanges them between
 * the ranks; sglk_ar_register_graph_buffers takes every rank's lists (handles[w]: count handles back to back,
 * offsets[w]: count offsets; the own entries are ignored), maps them and fills the records in
 * (register_graph_buffers, custom_all_reduce.hip:139-144).  The graph must not be replayed before that. */
SGLK_API int64_t sglk_ar_graph_buffer_count(int64_t fa);
SGLK_API int sglk_ar_get_graph_buffer_ipc_meta(int64_t fa, void* handles_out, int64_t* offsets_out, int64_t cap);
SGLK_API int sglk_ar_register_graph_buffers(int64_t fa, const void* const* handles,
                                            const int64_t* const* offsets, int64_t count);
/* out = sum over ranks of their registered `inp` (fp32 accumulation in rank order, one rounding:
 * every rank gets bit-identical results).  numel * element size must be a multiple of 16.
 * algo: 0 = choose as the reference does (custom_all_reduce_hip.cuh:543-553), 1 = one-stage,
 * 2 = two-stage.  (all_reduce_reg, custom_all_reduce.hip:89-95) */
SGLK_API int sglk_ar_all_reduce(int64_t fa, const void* inp, void* out, int64_t numel, int dtype,
                                int algo, void* stream);
/* Same for an unregistered input: it is first copied into the registered reg_buffer.
 * (all_reduce_unreg, custom_all_reduce.hip:97-117) */
SGLK_API int sglk_ar_all_reduce_unreg(int64_t fa, const void* inp, void* reg_buffer, void* out,
                                      int64_t numel, int dtype, int algo, void* stream);
/* out[r] = rank r's registered `inp` (bytes_per_rank bytes each, a multiple of 16), in rank order: the peer-to-peer
 * form of the all-gather GroupCoordinator.all_gather issues for vocab-parallel logits
 * (python/sglang/srt/distributed/parallel_state.py:489-540), usable inside HIP graphs. */
SGLK_API int sglk_ar_all_gather(int64_t fa, const void* inp, void* out, int64_t bytes_per_rank, void* stream);
/* Failure handling; no counterpart in the reference, whose barrier spins forever.  A barrier wait that sees no
 * peer within the time limit (default 30 s, or SGLK_AR_TIMEOUT_MS at init, or sglk_ar_set_timeout_ms) fills the
 * output with NaN instead of a sum and raises a host-pinned word.  sglk_ar_stuck reads that word (1 = failed;
 * does not synchronise: call it after the stream has been synchronised to judge the work before).  A failed
 * communicator poisons every later output and its host entry points return SGLK_ESTUCK; make a new one. */
SGLK_API int sglk_ar_stuck(int64_t fa);
SGLK_API int sglk_ar_set_timeout_ms(int64_t fa, int64_t ms);
/* IPC plumbing (hipIpcGetMemHandle / hipIpcOpenMemHandle / uncached allocation):
 * allocate_meta_buffer, get_meta_buffer_ipc_handle (custom_all_reduce.hip:146-170).
 * The handle names the allocation that holds dev_ptr; *offset_out is dev_ptr's offset in it. */
SGLK_API int sglk_ipc_handle_bytes(void);
SGLK_API int sglk_ipc_get_handle(void* dev_ptr, void* handle_out, int64_t* offset_out);
SGLK_API void* sglk_ipc_open_handle(const void* handle);
SGLK_API int sglk_ipc_close(void* mapped);
SGLK_API void* sglk_alloc_uncached(int64_t bytes); /* zero-filled */
SGLK_API int sglk_free_uncached(void* p);

/* ------------------------------------------------------------------ R1-R4: radix tree (host) -- */

/* Native prefix tree behind RadixCache (python/sglang/srt/mem_cache/radix_cache.py:77-653,
 * tree_node.py:28-79).  Host-side, no GPU.
 * Node handles are int64: (slot generation << 32) | slot index.  A node that leaves the tree (eviction, hit-rate
 * release, reset, an orphaned subtree once unlocked) is destroyed and its slot's generation bumped, so a handle
 * kept past its node -- a privacy verdict that arrives late, a request's last_node after a reset -- resolves
 * to nothing: setters return SGLK_EINVAL, lock calls return 0 and change nothing.  sglk_radix_root gives the
 * current root's handle (it changes at every reset).
 * compat: 1 reproduces the fork's behaviour bit for bit (fuzzy key match with up to 11 mismatches, private
 * root whose child an insert by another user overwrites, 1024-token chunks); 2 is the same with the slot leak
 * of that overwrite closed (the orphaned subtree's slots go back to the free queue); 0 is the exact-match
 * upstream behaviour. */
SGLK_API int64_t sglk_radix_create(int compat_fork);
SGLK_API void sglk_radix_destroy(int64_t tree);
SGLK_API void sglk_radix_reset(int64_t tree);
SGLK_API void sglk_radix_set_epoch(int64_t tree, int64_t epoch);
/* match_prefix (radix_cache.py:120-150,333-390).  user_id < 0: no user given.  Writes up to `cap`
 * matched slots into out_slots, returns the matched length; *last_node receives the node handle;
 * slots released by the entropy check (radix_cache.py:393-411) are appended to the free queue. */
SGLK_API int64_t sglk_radix_match_prefix(int64_t tree, const int32_t* key, int64_t key_len,
                                         int64_t user_id, int64_t* out_slots, int64_t cap,
                                         int64_t* last_node);
/* insert (radix_cache.py:152-158,501-564): returns the matched prefix length; ids of nodes created
 * (to be submitted to the privacy judge, radix_cache.py:544-548) are written to new_nodes. */
SGLK_API int64_t sglk_radix_insert(int64_t tree, const int32_t* key, const int64_t* slots,
                                   int64_t len, int64_t user_id, int64_t* new_nodes,
                                   int64_t new_cap, int64_t* n_new);
SGLK_API int64_t sglk_radix_inc_lock_ref(int64_t tree, int64_t node);
SGLK_API int64_t sglk_radix_dec_lock_ref(int64_t tree, int64_t node);
/* evict (radix_cache.py:249-283): frees leaves in epoch order until num_tokens are released; the
 * released slots are appended to the free queue. */
SGLK_API void sglk_radix_evict(int64_t tree, int64_t num_tokens);
/* drain the free queue (slots handed back to TokenToKVPoolAllocator.free) */
SGLK_API int64_t sglk_radix_take_freed(int64_t tree, int64_t* out, int64_t cap);
SGLK_API int64_t sglk_radix_evictable_size(int64_t tree);
SGLK_API int64_t sglk_radix_protected_size(int64_t tree);
SGLK_API int64_t sglk_radix_total_size(int64_t tree);
SGLK_API int64_t sglk_radix_root(int64_t tree);
/* nodes alive in host memory (reachable ones plus orphans still locked by a request) */
SGLK_API int64_t sglk_radix_live_nodes(int64_t tree);
/* SafeKV node state (tree_node.py:50-52; private_client.py:125-140 flips these asynchronously) */
SGLK_API int sglk_radix_node_set_privacy(int64_t tree, int64_t node, int is_private, int need_check);
SGLK_API int sglk_radix_node_get(int64_t tree, int64_t node, int64_t* fields /*[12]*/);
SGLK_API int64_t sglk_radix_node_key(int64_t tree, int64_t node, int32_t* key_out, int64_t* val_out,
                                     int64_t cap);
/* pre-order dump, children ordered by first token: per node 12 int64 fields (see sglk_radix_node_get) */
SGLK_API int64_t sglk_radix_dump(int64_t tree, int64_t* node_ids, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SGLK_H_ */
