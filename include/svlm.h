/* svlm.h -- C ABI of libsvlm_hip.so: the MI355X (gfx950) kernels of the streaming-VLM hot path.
 *
 * Drop-in boundary.  The reference (rahim-xelpmoc/streaming-vlm) is pure Python; its device work
 * is done by third-party wheels (transformers 4.52.4 modules, flash_attn 2.8, torch 2.7.1).  Each
 * entry point below replaces one of those call sites; the citation after "replaces:" is the
 * reference file:line (relative to /root/reference/src/streaming_vlm/inference unless noted) that
 * makes the call.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked [host]; bf16 data is raw uint16 bits;
 *   - the caller owns all buffers (torch allocator) and passes the HIP stream (hipStream_t as void*);
 *     kernels are stream-ordered, never synchronise, never allocate: safe under hipGraph capture;
 *   - return value: 0 = ok, negative errno-style code otherwise (SVLM_EINVAL bad argument,
 *     SVLM_ELAUNCH launch failure); svlm_last_error() returns the thread-local message;
 *     nothing ever throws across the boundary;
 *   - `len_dev` arguments: a device int32 holding the current KV length, so that one captured
 *     decode-step graph can be replayed while the cache grows; pass NULL to use host values.
 *
 * KV pool layout (allocated by the caller, one per stream):
 *     pool[layer][kv(0=K,1=V)][Hkv][n_slots][D]  bf16, keys UN-ROTATED (shrink mode,
 *     qwen2/language_forward.py:89-103); slot_of[i] = slot of logical token i.
 *     "k_planes"/"v_planes" below are &pool[layer][0] and &pool[layer][1].
 * RoPE table: rope_cs[i][0:D/2] = cos, [D/2:D] = sin of logical position i, bf16, mrope-section
 *     selected (built by svlm_mrope_table from the (3, L) position ids).
 */
#ifndef SVLM_H
#define SVLM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SVLM_OK 0
#define SVLM_EINVAL (-22)
#define SVLM_ELAUNCH (-5)

#define SVLM_ACT_NONE 0
#define SVLM_ACT_QUICK_GELU 1 /* x*sigmoid(1.702x), VisionMlp (qwen2/vision_forward.py:49) */
#define SVLM_ACT_GELU_ERF 2   /* nn.GELU(), PatchMerger (qwen2/vision_forward.py:80) */
#define SVLM_ACT_SILU 3
#define SVLM_ACT_SWIGLU 4      /* svlm_gemm_bf16 only: W = [gate rows; up rows] (2N x K), C (M x N) = bf16(bf16(silu(gate)) * up):
                                  Qwen2MLP's act_fn(gate_proj(x)) * up_proj(x) (qwen2/language_forward.py:201); bias, if any, is [gate; up] too; no residual */

int svlm_abi_version(void);
const char* svlm_last_error(void); /* [host] */
int svlm_device_cus(void);

/* C[M,N] = bf16(act(bf16(A[M,K].W[N,K]^T + bias)) + residual).  bias/residual may be NULL.
 * ws (may be NULL): fp32 scratch for split-K partial slabs, used when the tile grid alone cannot fill the chip.
 * replaces: nn.Linear / Conv3d-as-GEMM at qwen2/vision_forward.py:14,33,57,43-49,80 and
 * qwen2/language_forward.py:80-82,161,201 (prefill rows). */
int svlm_gemm_bf16(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                   void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes, void* stream);
/* svlm_gemm_bf16 followed by the norm of the GEMM's output rows: norm_b == NULL: XN[m,:] = norm_w * bf16(C[m,:] * rsqrt(mean(C^2) + eps))
 * (Qwen2RMSNorm); else XN[m,:] = bf16((C[m,:] - mean) * rstd * norm_w + norm_b) (nn.LayerNorm, the ViT's norm1 / norm2).
 * When the GEMM runs split-K the norm is computed inside the reduce launch; otherwise it is a second launch.
 * replaces: o_proj + residual -> post_attention_layernorm, down_proj + residual -> next input_layernorm
 * (qwen2/language_forward.py:161,195-200,183); ViT proj / fc2 + residual -> norm2 / next norm1 (qwen2/vision_forward.py:43-49). */
int svlm_gemm_bf16_norm(const void* A, int lda, const void* W, int ldw, const void* bias, const void* residual, int ldr,
                        void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes, const void* norm_w,
                        const void* norm_b, float eps, void* XN, int ldxn, void* stream);

/* y[N] = same epilogue for one row x[K] (decode step); y (bf16) and/or y_f32 (fp32 copy of the
 * bf16-rounded value: the `.float()` of streaming_generate_qwen.py:73) may be NULL.
 * replaces: the same Linear calls at T = 1 and lm_head on the last row (qwen2/model_forward.py:243). */
int svlm_gemv_bf16(const void* x, const void* W, int ldw, const void* bias, const void* residual, void* y, float* y_f32,
                   int N, int K, int act, void* stream);

/* Warm the 256 MiB Infinity Cache with [ptr, ptr+bytes) (16-B aligned) using n_wgs workgroups; a pure performance
 * hint for the weight stream of the next decode layer, launched on a side stream.  No reference counterpart. */
int svlm_prefetch(const void* ptr, long long bytes, int n_wgs, void* stream);

/* replaces: Qwen2RMSNorm (qwen2/language_forward.py:183,200,315). */
int svlm_rmsnorm(const void* x, const void* w, void* y, int rows, int cols, float eps, void* stream);
/* replaces: nn.LayerNorm of the ViT blocks and merger (qwen2/vision_forward.py:43-49,80). */
int svlm_layernorm(const void* x, const void* w, const void* b, void* y, int rows, int cols, float eps, void* stream);
/* y = bf16(a + b), n elements (residual adds, qwen2/language_forward.py:195,202). */
int svlm_add(const void* a, const void* b, void* y, long long n, void* stream);
/* h[r,:] = bf16(bf16(silu(gu[r,0:I])) * gu[r,I:2I])  (Qwen2MLP, qwen2/language_forward.py:201). */
int svlm_silu_mul(const void* gu, void* h, int rows, int inter, void* stream);
/* out[t] = idx[t] >= 0 ? table[idx[t]] : alt[-1-idx[t]]; idx read at idx[*idx_off + t] if idx_off != NULL.
 * replaces: embed_tokens + masked_scatter of the vision rows (qwen2/model_forward.py:34,62-69). */
int svlm_gather_rows(const void* table, const void* alt, const int* idx, const int* idx_off, void* out, int rows, int cols,
                     void* stream);

/* Frame ingest: uint8 frames (T,3,H,W) -> bf16 patches (ceil(T/temporal)*(H/patch)*(W/patch), 3*temporal*patch*patch),
 * rescaled by 1/255, normalised per channel, rows in merge-block-major order, a trailing odd frame repeated.
 * replaces: the HF video processor's rescale/normalize/patchify on the host (called at inference.py:390-395;
 * $TF/models/qwen2_vl/video_processing_qwen2_vl.py:247-270) -- SURVEY 8f-2. */
int svlm_patchify_u8(const void* frames, void* out, int T, int H, int W, int patch, int temporal, int merge,
                     float m0, float m1, float m2, float s0, float s1, float s2, void* stream);
/* Frame ingest, resize: uint8 planes (planes = T*C, H, W) -> uint8 (planes, h, w) by the separable antialiased bicubic filter
 * of torch's interpolate(mode="bicubic", antialias=True, align_corners=False) in fp32 -- width pass, height pass, clamp to
 * [0, 255], round half to even.  svlm_resize_aa_tables is HOST arithmetic (callable without a GPU): the tap tables of one axis
 * (first tap, tap count, normalised weights with row stride wt_stride >= K); it returns K and, with NULL pointers, only sizes
 * the buffers.  The caller uploads the two axes' tables; ws >= svlm_resize_ws_bytes(planes, H, w) holds the fp32 width pass.
 * replaces: _spatial_resize_video = torchvision resize(BICUBIC, antialias=True) on the decoded clip
 * (livecc_utils/src/livecc_utils/video_process_patch.py:134-153) -- SURVEY 8f-2. */
int svlm_resize_aa_tables(int in_size, int out_size, int* xmin, int* xsize, float* wt, int wt_stride);
long long svlm_resize_ws_bytes(int planes, int H, int w);
int svlm_resize_bicubic_aa_u8(const void* src, void* dst, int planes, int H, int W, int h, int w, const int* xmin, const int* xsize,
                              const float* wt_x, int Kx, const int* ymin, const int* ysize, const float* wt_y, int Ky, void* ws,
                              long long ws_bytes, void* stream);

/* ---- FP8 (OCP e4m3) ViT GEMMs of BASELINE configs[4] ("fp8 MFMA ViT path").  The reference has no fp8 path; the recipe is the
 * build's: activations get one dynamic fp32 scale per ROW (max|row| / 448), weights one static scale per OUTPUT CHANNEL, both
 * sides round-to-nearest-even to e4m3, the products accumulate in fp32 on v_mfma_f32_16x16x32_fp8_fp8 and the scales are applied
 * before the bf16 epilogue.  oracle/model.py:linear_fp8 restates it; parity is HIP vs that oracle leg.
 * svlm_quant_rows_fp8: x bf16 (rows, ldx) -> q fp8 (rows, ldq) + scale fp32 (rows).
 * svlm_gemm_fp8: C = epi((A8 . W8^T) * a_scale[m] * w_scale[n]) with the epilogue (bias, act, residual) and the optional fused
 * norm (norm_w / norm_b / XN as svlm_gemm_bf16_norm) of the bf16 GEMM; K % 128 == 0; ws = fp32 split-K scratch.
 * replaces: the ViT Linear calls at qwen2/vision_forward.py:14,33,43-49 and the merger MLP (:80) in the fp8 configuration. */
int svlm_quant_rows_fp8(const void* x, int ldx, void* q, int ldq, float* scale, int rows, int cols, void* stream);
int svlm_gemm_fp8(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                  const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                  const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* stream);
/* svlm_gemm_fp8 whose fused norm ALSO leaves the normalised rows as the next GEMM's operand: XN8 (e4m3, rows of ldxn8 bytes) and
 * xn_scale[M], exactly what svlm_quant_rows_fp8 would make of XN -- the quantiser launch between two Linears of the fp8 tower
 * disappears (LayerNorm rows: inside the split-K reduce; RMSNorm rows: by the stand-alone quantiser behind it).
 * replaces: the same Linears as svlm_gemm_fp8 (qwen2/vision_forward.py:43-49 in the fp8 configuration of BASELINE configs[4]). */
int svlm_gemm_fp8_normq(const void* A8, int lda, const float* a_scale, const void* W8, int ldw, const float* w_scale, const void* bias,
                        const void* residual, int ldr, void* C, int ldc, int M, int N, int K, int act, void* ws, long long ws_bytes,
                        const void* norm_w, const void* norm_b, float eps, void* XN, int ldxn, void* XN8, int ldxn8, float* xn_scale,
                        void* stream);

/* In-place 2-D rope on the q and k parts of the fused ViT qkv buffer (N,3,H,d); cosT/sinT fp32 (N,d/2).
 * replaces: apply_rotary_pos_emb_vision (qwen2/vision_forward.py:27). */
int svlm_vit_rope(void* qkv, const float* cosT, const float* sinT, int N, int H, int d, void* stream);
/* Block-diagonal non-causal attention over n_seq sequences of seq_len rows; out (N, H*d).
 * replaces: flash_attn_varlen_func (qwen2/vision_forward.py:30). */
int svlm_vit_attn(const void* qkv, void* out, int n_seq, int seq_len, int H, int d, float scale, void* stream);

/* rope_cs rows [start, start+count) from position ids pos3 (int32) or posf3 (fp32, Qwen2.5), each
 * (3, pos_stride); inv_freq fp32 (D/2) computed by the host exactly as torch does.
 * replaces: Qwen2VLRotaryEmbedding.forward + mrope section select (qwen2/language_forward.py:271,43-60). */
int svlm_mrope_table(const int* pos3, const float* posf3, int pos_stride, const float* inv_freq, void* rope_cs, int start,
                     int count, int head_dim, int sec_t, int sec_h, int sec_w, void* stream);

/* M-RoPE position ids of a whole id sequence, computed on the device: ids (L) and grids (n_grids, 3) [t, h, w in
 * patches] int32 in device memory -> pos3 (3, stride) int32, or posf3 fp32 with the Qwen2.5 temporal scaling
 * ((t * second_per_grid_t) * tokens_per_second, then + text_len + start, in the reference's order); rows [L, L + n_extra) continue
 * the trailing text run (the tokens about to be generated).  ws >= svlm_rope_index_ws_bytes(L, n_grids); its first int receives
 * 0 or the reference's failure (2: more spans than grid rows, 3: span without video tokens, 4: span past the end).
 * replaces: get_rope_index, inference/qwen2/pos_emb.py:69-133 (Qwen2.5: inference/qwen2_5/pos_emb.py:107-160), recomputed there
 * in Python on every forward (qwen2/model_forward.py:119-126). */
long long svlm_rope_index_ws_bytes(int max_len, int max_spans);
int svlm_rope_index(const int* ids, int L, const int* grids, int n_grids, int merge, int video_token_id, int vision_start_token_id,
                    int* pos3, float* posf3, int stride, float second_per_grid_t, float tokens_per_second, int n_extra, void* ws,
                    long long ws_bytes, void* stream);

/* Span finder + eviction policy on the device: one single-workgroup kernel over the device copy of the ids computes what
 * the reference computes in Python per chunk -- every get_qwen_range (utils/get_qwen_range.py:15-86) the policy asks for, the policy
 * itself (policy 1: process_past_kv, inference.py:87-172, round = chunk index; policy 0: BASELINE sink/window with the cut end
 * snapped to <|vision_end|>) and the id edits (prune_id_and_kv_cache :50-61, resort_id_and_kv :100-108).
 * ws (>= svlm_evict_plan_ws_bytes(L)) starts with int out[4 + 64]: status (0 ok, 1 a span the policy needs is missing, 2 more than
 * 16 ops), n_ops, new length, result buffer (0: `ids` itself, 1: the int array at ws + (4 + 64) * 4), then n_ops x (1 = prune | 2 =
 * move, start, end, dst): the eviction indices the host applies to the KV pool's slot table.  text_sink / text_sliding_window < 0
 * mean None; tokens = HOST array {<|im_start|>, <|im_end|>, user, assistant, <|vision_start|>, <|vision_end|>, <|video_pad|>, "\n",
 * previous, " text", Time} (get_qwen_range.py:2-13). */
long long svlm_evict_plan_ws_bytes(int max_len);
int svlm_evict_plan(int* ids, int L, int policy, int round, int text_round, int visual_round, int text_sink, int text_sliding_window,
                    int assistant_start_bias, int assistant_end_bias, int sink, int window, int kv_len, const int* tokens, void* ws,
                    long long ws_bytes, void* stream);

/* Append T un-rotated K/V rows of one layer at logical rows base..base+T-1 (base = *len_dev or start).
 * replaces: StreamingCache.update = torch.cat (generate/streaming_cache.py:72-73). */
int svlm_kv_append(const void* k_new, int k_stride, const void* v_new, int v_stride, void* k_planes, void* v_planes,
                   const int* slot_of, const int* len_dev, int start, int T, int Hkv, int D, int n_slots, void* stream);
/* In-place defragmentation: for every plane, row src[i] -> row dst[i] (dst = free slots).
 * replaces (together with host edits of slot_of): index_select eviction (inference.py:54-59), the
 * 4-slice torch.cat row move (inference.py:106-107) and .contiguous() (inference.py:66-67). */
int svlm_kv_move_rows(void* pool, long long n_planes, int n_slots, int D, const int* src, const int* dst, int n, void* stream);
/* Dense (Hkv, L, D) copy of one K or V region in logical order (StreamingCache-compatible views). */
int svlm_kv_gather(const void* planes, const int* slot_of, void* out, int L, int Hkv, int D, int n_slots, void* stream);

/* Decode-step attention: q (Hq, D) un-rotated; length = (*len_dev if len_dev else 0) + len_add counts the
 * row just appended; ws >= svlm_decode_attn_ws_bytes(Hq, max_len, chunk); out (Hq, D).
 * replaces: post-cache M-RoPE + repeat_kv + _flash_attention_forward at q_len = 1
 * (qwen2/language_forward.py:103,107-108,148-158). */
long long svlm_decode_attn_ws_bytes(int Hq, int max_len, int chunk);
int svlm_decode_attn_ropeload(const void* q, const void* k_planes, const void* v_planes, const int* slot_of, const void* rope_cs,
                              const int* len_dev, int len_add, void* out, void* ws, int Hq, int Hkv, int D, int n_slots,
                              int max_len, int chunk, float scale, void* stream);
/* The same attention with the layer's LINEAR PLANES beside the pool.  The reference rotates EVERY cached key in EVERY forward
 * (qwen2/language_forward.py:55-63, called at :103); between two evictions the positions of the cached rows do not change, so the
 * rotated keys the prefill of a chunk produces anyway are kept (svlm_prefill_attn_ropeload_lin) and each decode step streams them:
 *   k_lin (Hkv, lin_rows/16, 4, 64, 8) bf16  16-key tiles of ROTATED keys in logical order; inside a tile chunk c (8 values) of key r
 *                                            sits at [c >> 2][(c & 3) * 16 + r][8], the operand layout of the decode kernels
 *   v_lin (Hkv, lin_rows, 128) bf16          values in logical order
 *   lin_len_dev[2] = {R, F}                  rows [0, R) of both are valid with ROTATED keys; F != 0: the rows appended since the
 *                                            prefill follow at their logical rows with UN-rotated keys (svlm_dec_qkv_lin) and are
 *                                            rotated while they are staged; F == 0: rows >= R come from the pool as in
 *                                            svlm_decode_attn_ropeload
 * lin_rows % 16 == 0, lin_rows >= max_len; all three NULL = svlm_decode_attn_ropeload.  Same bits every way. */
int svlm_decode_attn_lin(const void* q, const void* k_planes, const void* v_planes, const int* slot_of, const void* rope_cs,
                         const int* len_dev, int len_add, const void* k_lin, const void* v_lin, int lin_rows, const int* lin_len_dev,
                         void* out, void* ws, int Hq, int Hkv, int D, int n_slots, int max_len, int chunk, float scale, void* stream);
/* Prefill attention: q (T, q_stride) un-rotated rows for logical positions L-T..L-1, causal bottom-right aligned.
 * Their un-rotated K/V rows are either already in the pool (k_new = v_new = NULL, after svlm_kv_append) or handed over as
 * k_new / v_new (T, kv_new_stride) -- e.g. column slices of the fused QKV projection -- and APPENDED to their slots by the
 * same launch that rotates the keys (StreamingCache.update, generate/streaming_cache.py:72-73).
 * out (T, o_stride); ws >= svlm_prefill_attn_ws_bytes(T, L, Hq, Hkv)
 * holds the rotated queries, this layer's rotated keys / gathered values in logical order and, when the
 * query tiles alone cannot fill the chip, the fp32 (O, m, l) partials of up to 8 key splits.
 * replaces: same lines at q_len = T. */
long long svlm_prefill_attn_ws_bytes(int T, int L, int Hq, int Hkv);
int svlm_prefill_attn_ropeload(const void* q, int q_stride, const void* k_new, const void* v_new, int kv_new_stride,
                               void* k_planes, void* v_planes, const int* slot_of, const void* rope_cs, void* out, int o_stride,
                               int T, int L, int Hq, int Hkv, int D, int n_slots, float scale, void* ws, long long ws_bytes,
                               void* stream);
/* ... which also leaves rows [0, L) of the layer's linear planes (svlm_decode_attn_lin) behind and sets lin_len_dev[0..1] = {L, 1}
 * (k_lin, v_lin, lin_len_dev together or all NULL; lin_rows % 16 == 0, lin_rows >= L). */
int svlm_prefill_attn_ropeload_lin(const void* q, int q_stride, const void* k_new, const void* v_new, int kv_new_stride,
                                   void* k_planes, void* v_planes, const int* slot_of, const void* rope_cs, void* out, int o_stride,
                                   int T, int L, int Hq, int Hkv, int D, int n_slots, float scale, void* ws, long long ws_bytes,
                                   void* k_lin, void* v_lin, int lin_rows, int* lin_len_dev, void* stream);

/* seen[id] = 1 for ids[0..n).  (input to the repetition penalty) */
int svlm_mark_seen(const int* ids, int n, void* seen, int V, void* stream);
/* Repetition penalty + argmax (lowest index on ties) + device-side token feedback:
 * tok_buf[state[1]+1] = token; state[1] += 1; state[0] += advance_kv; seen[token] = 1.
 * replaces: logits processors + argmax + cat (generate/streaming_generate_qwen.py:75,99,104). */
long long svlm_argmax_ws_bytes(void);
int svlm_penalty_argmax(const float* logits, int V, void* seen, float penalty, const int* suppress, int n_suppress, int* tok_buf,
                        int* state, int advance_kv, void* ws, void* stream);

/* The reference's DEFAULT token choice (do_sample=True at inference.py:446; generate/streaming_generate_qwen.py:75,95-97 with the
 * warpers HF builds from the generation config): repetition penalty -> temperature -> top-k (0 = off; every score >= the k-th
 * largest survives) -> top-p (1 = off) -> softmax -> one multinomial draw, then the same device-side token feedback as
 * svlm_penalty_argmax.  rng = {seed lo, seed hi} in DEVICE memory (Philox4x32-10, counted by the generated-token index
 * state[1]+1 and the vocabulary index); top_k == 1 is the argmax; top_k == 0 && top_p == 1 is a Gumbel-max draw inside the
 * argmax kernels; 1 < top_k < 2048 runs one single-workgroup select / sort / cut / inverse-CDF kernel; top_p alone (or a wider
 * k) runs the threshold form -- radix descent by count and by probability mass, then a Gumbel-max over the survivors -- which is
 * exact for a nucleus of any size.  ws >= svlm_argmax_ws_bytes(). */
int svlm_penalty_sample(const float* logits, int V, void* seen, float penalty, const int* suppress, int n_suppress, float temperature,
                        int top_k, float top_p, const unsigned* rng, int* tok_buf, int* state, int advance_kv, void* ws, void* stream);

/* ---- fused decode-step (T = 1) kernels: the per-layer small ops folded into the weight-streaming GEMVs ---- */
/* RMSNorm(x; ln_w) -> W x + bias -> q_out (qd) and the new token's K/V rows written straight into the pool slot
 * slot_of[*len_dev or len_host].  replaces: qwen2/language_forward.py:183,80-82 + generate/streaming_cache.py:72-73. */
int svlm_dec_qkv(const void* x, const void* ln_w, float eps, const void* W, int ldw, const void* bias, void* q_out, void* k_planes,
                 void* v_planes, const int* slot_of, const int* len_dev, int len_host, int K, int qd, int kd, int D, int n_slots,
                 void* stream);
/* ... and ALSO into the layer's linear planes (svlm_decode_attn_lin) at logical row *len_dev or len_host: V as it is, K un-rotated
 * in its tile position (k_lin, v_lin together or both NULL; the row must be below lin_rows). */
int svlm_dec_qkv_lin(const void* x, const void* ln_w, float eps, const void* W, int ldw, const void* bias, void* q_out, void* k_planes,
                     void* v_planes, const int* slot_of, const int* len_dev, int len_host, int K, int qd, int kd, int D, int n_slots,
                     void* k_lin, void* v_lin, int lin_rows, void* stream);
/* RMSNorm(x; ln_w) -> h[n] = silu(Wg[n] x) * (Wu[n] x), W = [gate(I) | up(I)] rows.  replaces: :200-201 (Qwen2MLP). */
int svlm_dec_gate_up(const void* x, const void* ln_w, float eps, const void* W, int ldw, void* h, int I, int K, void* stream);

/* ---- persistent decode-layer tail: o_proj + residual -> RMSNorm -> gate/up + SwiGLU -> down_proj + residual -> (next layer's)
 * RMSNorm -> QKV + bias + KV append, as ONE launch of one 16-wave workgroup per CU (csrc/dec_tail.hip).  Every consumer wave requests
 * ALL its gate/up and down_proj rows in its first microsecond (the whole layer's tail, 93.6 MB on Qwen2-VL-2B, is in flight in the
 * register files at once), so HBM streams the layer without a pause while the three all-to-all seams inside the launch -- 8-byte
 * {tag, 2 x bf16} granule hand-offs swept by one gatherer wave per workgroup -- go by.  Same rounding points as svlm_gemv_bf16
 * (o_proj, down_proj), svlm_dec_gate_up and svlm_dec_qkv, which it replaces inside a decode step.
 * replaces: the per-layer module calls of qwen2/language_forward.py:161 (o_proj), :196-202 (residual, post_attention_layernorm, Qwen2MLP,
 * residual) and the next layer's :183,80-82 (input_layernorm, q/k/v_proj) + generate/streaming_cache.py:72-73 (cache append), i.e.
 * the body of the per-layer loop at :278 between two attention calls.
 *   svlm_dec_tail_supported: 1 when the layer geometry has a build at `grid` workgroups (its weights must fit the register file:
 *       Qwen2-VL-2B class and small test widths), 0 otherwise (use the per-op entry points); host arithmetic.
 *   ws: svlm_dec_tail_ws_bytes(H, I, n_layers) bytes = [256-B status block][one granule block per layer]; int status = ((int*)ws)[0] is
 *       sticky: non-zero once any gatherer gave up its (bounded) spin -- results of that step are then invalid; the caller checks it once
 *       per chunk and clears it.  svlm_dec_tail_reset zeroes the granule blocks (NOT the status) and must run once before the tails of
 *       every decode step (a memset node at the head of the step's graph).
 *   attn [qd], x [H] (in: residual stream, out: the layer's output), weights as in the per-op entry points.
 *   ln1_next == NULL: last layer, no QKV phase (q_out / planes / slot_of unused).
 *   grid: workgroups to launch, 0 = one per CU; never more than the device has CUs (every workgroup waits for every other one's
 *       outputs, so all must be resident together); results do not depend on where they land.
 *   stamps: NULL, or [grid][2][16] uint64 wall-clock stamps of the gatherer's and the first consumer's barrier passes (a profiling aid of
 *       tools/dec_tail_bench.py). */
int svlm_dec_tail_supported(int H, int I, int qd, int kd, int grid); /* [host] */
long long svlm_dec_tail_ws_bytes(int H, int I, int n_layers); /* [host] */
int svlm_dec_tail_reset(void* ws, int H, int I, int n_layers, void* stream);
int svlm_dec_tail(const void* attn, void* x, const void* o_w, int ld_o, const void* ln2, const void* gu_w, int ld_gu, const void* down_w,
                  int ld_down, const void* ln1_next, const void* qkv_w_next, int ld_qkv, const void* qkv_b_next, void* q_out,
                  void* k_planes_next, void* v_planes_next, const int* slot_of, const int* len_dev, int len_host, int H, int I, int qd, int kd,
                  int D, int n_slots, float eps, void* ws, int layer, int n_layers, int grid, void* stamps, void* stream);
/* final RMSNorm -> last-row logits (fp32 copy of the bf16 value) -> penalty / suppression -> per-workgroup argmax
 * candidates in ws (>= svlm_dec_lm_head_ws_bytes(V)); svlm_argmax_finish picks the winner and feeds it back
 * (same state protocol as svlm_penalty_argmax).  replaces: qwen2/language_forward.py:315, qwen2/model_forward.py:243,
 * generate/streaming_generate_qwen.py:73-104. */
long long svlm_dec_lm_head_ws_bytes(int V);
int svlm_dec_lm_head(const void* x, const void* ln_w, float eps, const void* W, int ldw, float* logits, const void* seen, float penalty,
                     const int* suppress, int n_suppress, void* ws, int V, int K, void* stream);
int svlm_argmax_finish(const void* ws, int V, void* seen, int* tok_buf, int* state, int advance_kv, void* stream);
/* svlm_dec_lm_head with plain temperature sampling folded in: the candidates are argmax(score / T + Gumbel noise), an exact
 * draw from softmax(score / T); rng / state as svlm_penalty_sample.  svlm_argmax_finish completes it.
 * replaces: generate/streaming_generate_qwen.py:95-97 (softmax + torch.multinomial) on decode steps. */
int svlm_dec_lm_head_sample(const void* x, const void* ln_w, float eps, const void* W, int ldw, float* logits, const void* seen,
                            float penalty, const int* suppress, int n_suppress, void* ws, int V, int K, float temperature,
                            const unsigned* rng, const int* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVLM_H */
