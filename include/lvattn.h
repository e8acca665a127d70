/* lvattn.h — C ABI of the MI355X (gfx950) attention hot path for Long-VITA.
 *
 * The reference has no FFI of its own: its "operator API" on this path is a set
 * of Python call sites into kernels that live in torch_npu / MindSpeed.  Every
 * entry point below names the reference call site it stands in for.  All
 * pointers are raw device pointers unless the name ends in _host; sizes and
 * strides are in ELEMENTS (bf16) unless stated; `stream` is a hipStream_t passed
 * as void*.  Return value: 0 on success, a negative LV_E* code otherwise, with a
 * message retrievable through lvattn_last_error().  No entry point allocates,
 * frees or synchronises: callers pass every buffer (graph-capture friendly).
 *
 * Block descriptors
 * -----------------
 * The attention kernels are driven by a device array of 8-int32 descriptors, one
 * per (query-block x key-range) work item, so that the same kernel serves packed
 * variable-length causal attention, dense ViT attention and the sub-blocks of the
 * zig-zag ring:
 *   [0] q_row0   first query row of the block (row index into the q / out tensors)
 *   [1] q_rows   valid query rows in the block (1 .. LV_FWD_BLOCK_M)
 *   [2] k_row0   first key row the block may attend to (row index into k / v)
 *   [3] k_len    number of key rows from k_row0 the block may attend to
 *   [4] shift    causal rule: key k_row0+j visible to query q_row0+i iff j <= i + shift
 *                (ignored by the non-causal kernels)
 *   [5] out_delta  results (out / lse rows; dq rows) go to row q_row0 + out_delta + i instead of q_row0 + i: 0 for the
 *                builders above the "split sweeps" section, a workspace row offset for the split tables
 *   [6] band + 1  0: plain causal rule.  b + 1: key k_row0+j visible to query q_row0+i iff i + shift - b <= j <= i + shift
 *                (a query sees itself and the b keys before it: pre_tockens / --sliding-window,
 *                dot_product_attention.py:403-406, 462; written by the *_band builders, read by launches with causal == 2)
 *   [7] reserved, must be 0
 * The backward kernels use the transposed descriptor (one per key block):
 *   [0] k_row0  [1] k_rows (1 .. LV_BWD_BLOCK_N)  [2] q_row0  [3] q_len
 *   [4] shift   key k_row0+j visible to query q_row0+i iff j <= i + shift (q_row0 is the first
 *               32-aligned query row that sees any key of the block)
 *   [5] out_delta  dk / dv rows go to row k_row0 + out_delta + j
 *   [6] band + 1   as above
 */
#ifndef LVATTN_H
#define LVATTN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LV_OK 0
#define LV_EINVAL (-1)   /* bad argument (shape, stride, alignment, null pointer) */
#define LV_ELAUNCH (-2)  /* HIP reported a launch error */
#define LV_ESIZE (-3)    /* output buffer too small */

#define LV_FWD_BLOCK_M 256 /* query rows per forward work item  */
#define LV_BWD_BLOCK_N 128 /* key rows per dK/dV work item         */
#define LV_DESC_INTS 8

/* Library identification / diagnostics. */
int lvattn_abi_version(void);
/* Query rows per forward work item (LV_FWD_BLOCK_M unless the environment selects the 4-wave
 * lockstep kernel with LVATTN_FWD_KERNEL=lockstep LVATTN_FWD_WAVES=4: then 128). */
int lvattn_fwd_block_m(void);
const char* lvattn_last_error(void);

/* Diagnostic, no counterpart in the reference: the shader clock the chip holds under a dense bf16 MFMA loop (the 2.5
 * PFLOP/s peak assumes 2.4 GHz).  n_workgroups x 4 waves run `iters` x 4 v_mfma_f32_32x32x16_bf16 on the 1024 bf16
 * values at `seed` (random data: zeros clock higher) and write, per wave, {shader-clock ticks, ticks of the constant
 * 100 MHz counter} to stamps[2 * wave ..] (uint64); sink [n_workgroups * 4] floats keeps the arithmetic alive.
 * clock in MHz = 100 * stamps[2w] / stamps[2w + 1]. */
int lv_clock_probe(const void* seed, uint64_t* stamps, float* sink, int n_workgroups, int iters, void* stream);

/* ---- host-side block builders (pure CPU, no HIP calls) -------------------
 * cu_seqlens_host = [0] + the reference's `actual_seq_len` cumulative end offsets
 * (lcvlm_modellink/core/transformer/dot_product_attention.py:333,363-364;
 * produced by cognitron_vl/data/dataset_qwen2.py:166,209-219 and
 * lcvlm_modellink/legacy/data/data_samplers.py:45-48).
 * Writes up to max_blocks descriptors ordered heaviest-first; returns the count,
 * or LV_ESIZE if max_blocks is too small (needed: sum ceil(L_i / block)). */
int lv_build_fwd_blocks(const int32_t* cu_seqlens_host, int n_seg, int causal,
                        int32_t* desc_host, int max_blocks);
int lv_build_bwd_blocks(const int32_t* cu_seqlens_host, int n_seg, int causal,
                        int32_t* desc_host, int max_blocks);
/* Banded causal attention (the fused op's band modes: sparse_mode 0 / 1 / 4 with pre_tockens smaller than a segment,
 * lcvlm_modellink/core/transformer/dot_product_attention.py:403-406, 454-467; --sliding-window sets pre_tockens): a query
 * sees itself and the `band` keys before it inside its segment.  The sweep of a work item is trimmed to the band.  Launch
 * lvattn_fwd / lvattn_bwd on these tables with causal == 2 (the kernels that read the band are then selected). */
int lv_build_fwd_blocks_band(const int32_t* cu_seqlens_host, int n_seg, int band, int32_t* desc_host, int max_blocks);
int lv_build_bwd_blocks_band(const int32_t* cu_seqlens_host, int n_seg, int band, int32_t* desc_host, int max_blocks);
int lv_build_fwd_blocks_rect_band(int q_row0, int q_len, int k_row0, int k_len, int shift, int band, int32_t* desc_host, int max_blocks);
/* One rectangular (q rows x k rows) region, e.g. a ring step: q_row0..+q_len
 * against k_row0..+k_len, with `shift` as above when causal. */
int lv_build_fwd_blocks_rect(int q_row0, int q_len, int k_row0, int k_len, int causal, int shift,
                             int32_t* desc_host, int max_blocks);
int lv_build_bwd_blocks_rect(int q_row0, int q_len, int k_row0, int k_len, int causal, int shift,
                             int32_t* desc_host, int max_blocks);

/* ---- attention forward ------------------------------------------------------
 * Replaces torch_npu.npu_fusion_attention(...)[0] at
 *   dot_product_attention.py:352-365 (TND, packed varlen causal, sparse_mode=2),
 *   dot_product_attention.py:314-325 (SBH, ViT, no mask; causal=0),
 *   dot_product_attention.py:454-467 (generic causal),
 * and the per-step block attention inside MindSpeed's ringattn_context_parallel
 * (call site dot_product_attention.py:70-71).
 * q [rows, hq, d], k/v [rows, hkv, d], out [rows, hq, d] bf16 with explicit row and
 * head strides (last dim contiguous, 16-byte aligned rows); lse fp32 [hq, lse_stride]
 * natural-log log-sum-exp of the scaled scores (-inf for rows with no visible key,
 * whose output is 0).  d in {64, 128}; hq % hkv == 0.
 * q_rows_total / k_rows_total bound the row indices any descriptor may touch. */
int lvattn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
               const int32_t* desc_dev, int n_blocks,
               int q_rows_total, int k_rows_total, int hq, int hkv, int d,
               int64_t q_row_stride, int64_t q_head_stride,
               int64_t k_row_stride, int64_t k_head_stride,
               int64_t v_row_stride, int64_t v_head_stride,
               int64_t o_row_stride, int64_t o_head_stride,
               int64_t lse_stride, float softmax_scale, int causal, void* stream);

/* lvattn_fwd for ONE dense segment without descriptors: q_rows query rows against k_rows keys, every head.  Replaces
 * the same call sites where the reference has no packing: the vision tower's dot_product_attention.py:314-325
 * (q = k = v rows [1025], batch x heads as the head axis, causal = 0) and a single-sequence causal prefill
 * (:454-467).  With causal = 1 the rule is aligned to the END of both ranges (row i sees keys
 * j <= i + k_rows - q_rows; k_rows >= q_rows).  The kernels number their work items themselves, and the workgroup
 * size follows the row count (the tower's 1025 rows run as 9 workgroups of 128 rows instead of 4 of 256 plus one
 * that holds 8 waves for a single row).  Same tensors, strides, lse and errors as lvattn_fwd. */
int lvattn_fwd_dense(const void* q, const void* k, const void* v, void* out, float* lse,
                     int q_rows, int k_rows, int hq, int hkv, int d,
                     int64_t q_row_stride, int64_t q_head_stride,
                     int64_t k_row_stride, int64_t k_head_stride,
                     int64_t v_row_stride, int64_t v_head_stride,
                     int64_t o_row_stride, int64_t o_head_stride,
                     int64_t lse_stride, float softmax_scale, int causal, void* stream);

/* lvattn_fwd with the caller's knowledge of the work items: mean_keys_per_item = keys swept per descriptor on average
 * (lv_mean_keys_per_item over the host copy of the table; < 0 = unknown, which makes this lvattn_fwd).  Same call sites as
 * lvattn_fwd; the figure only steers the choice between the two head-dim-128 forward kernels (long sweeps: one wave per
 * SIMD; short ones, as in a launch of many short packed segments at dot_product_attention.py:352-365: the ping-pong
 * kernel, 5-20 % faster there), never the result beyond the kernels' rounding. */
int lvattn_fwd_hinted(const void* q, const void* k, const void* v, void* out, float* lse,
                      const int32_t* desc_dev, int n_blocks,
                      int q_rows_total, int k_rows_total, int hq, int hkv, int d,
                      int64_t q_row_stride, int64_t q_head_stride,
                      int64_t k_row_stride, int64_t k_head_stride,
                      int64_t v_row_stride, int64_t v_head_stride,
                      int64_t o_row_stride, int64_t o_head_stride,
                      int64_t lse_stride, float softmax_scale, int causal, int mean_keys_per_item, void* stream);
/* Host side, pure CPU: the mean key count per descriptor of a table the builders above wrote (-1 for an empty table). */
int lv_mean_keys_per_item(const int32_t* desc_host, int n_blocks);

/* TEST HOOKS, NOT FOR PRODUCTION (no counterpart in the reference).  The two selectors below set process-global,
 * unsynchronised state: they exist so that the test-suite can force every kernel variant through the same entry
 * points and so that A/B timing runs can compare variants inside one process.  A deployment leaves them alone (the
 * library's own choice is the measured-best one per launch shape) and must not call them while launches are in
 * flight on another thread.
 * lvattn_select_fwd_kernel: the forward kernel for later lvattn_fwd calls of this process ("default", "a4", "pp",
 * "pp16", "lockstep"; the same names as the LVATTN_FWD_KERNEL environment variable, which is read on first use
 * otherwise). */
int lvattn_select_fwd_kernel(const char* name);
/* The same for the backward: dq in {"default", "pp", "lockstep"}, dkdv in {"default" (the gap-scheduled one-wave kernel
 * at head dim 128, the role-split kernel at head dim 64 and for banded descriptors), "g", "rs" (role-split, two units per
 * iteration), "rs1" (one unit), "single"} -- LVATTN_DQ_KERNEL / LVATTN_DKDV_KERNEL / LVATTN_DKDV_UNITS otherwise. */
int lvattn_select_bwd_kernels(const char* dq, const char* dkdv);

/* ---- split sweeps: filling the chip at small head counts --------------------------
 * The reference runs tensor parallel 8 (scripts/modellink/qwen25/finetune_qwen25_14b_intern_300m_ptd_tp8pp1_stage1.sh:101):
 * each device sees 5 query heads and 1 KV head at the TND call (lcvlm_modellink/core/transformer/dot_product_attention.py:
 * 352-365).  A packed batch of a few thousand tokens is then a few dozen work items (one 256-row query block, or one
 * 128-key block, of one head) on 256 compute units.  These entry points cut the sweeps into pieces: every (block, piece)
 * pair is a work item of its own whose result goes to workspace rows of its own (descriptor word 5, "out_delta": results
 * are written to row a_row0 + out_delta + i), and a combine kernel merges a block's pieces in a fixed order (bitwise
 * reproducible, no atomics).  No counterpart in the reference: the NPU op schedules its own cores.
 *   lv_plan_fwd_split / lv_plan_bwd_split   host, pure CPU: the piece length (keys per piece for the forward and dQ, query rows
 *       per piece for dK/dV) a simulated heaviest-first schedule on n_cu compute units likes best, or 0 when the launch has
 *       enough work items (2 x n_cu) or splitting would gain under a tenth.
 *   lv_build_fwd_blocks_split / lv_build_bwd_blocks_split   host: the descriptor table (heaviest first; returns the count;
 *       desc_host == NULL only counts) and the combine table comb_host [n_comb][4] = (out_row0, rows, ws_row0, n_pieces): piece p
 *       of the block is at workspace rows ws_row0 + p * B .. (B = 256 forward / dQ, 128 dK/dV); *ws_rows = workspace rows needed.
 *       Every block goes through the workspace, also those with one piece.
 *   lvattn_combine_lse   out / lse <- log-sum-exp merge of the pieces' (out bf16 [ws_rows, hq, d] contiguous, lse [hq, ws_lse_stride])
 *       written by lvattn_fwd with the split table (pass the workspace as its out / lse).
 *   lvattn_combine_sum   out (bf16) <- sum of the pieces' fp32 partials ws [ws_rows, heads, d] contiguous, written by lvattn_bwd
 *       with accum_f32 = 1 into a ZEROED workspace (dq with the forward's split table, dk / dv with the backward's). */
int lv_plan_fwd_split(const int32_t* cu_seqlens_host, int n_seg, int causal, int heads, int n_cu);
int lv_plan_bwd_split(const int32_t* cu_seqlens_host, int n_seg, int causal, int kv_heads, int group, int n_cu);
int lv_build_fwd_blocks_split(const int32_t* cu_seqlens_host, int n_seg, int causal, int piece_keys, int32_t* desc_host, int max_blocks,
                              int32_t* comb_host, int max_comb, int* n_comb, int* ws_rows);
int lv_build_bwd_blocks_split(const int32_t* cu_seqlens_host, int n_seg, int causal, int piece_rows, int32_t* desc_host, int max_blocks,
                              int32_t* comb_host, int max_comb, int* n_comb, int* ws_rows);
int lvattn_combine_lse(const void* ws_out, const float* ws_lse, void* out, float* lse, const int32_t* comb_dev, int n_comb,
                       int hq, int d, int64_t ws_lse_stride, int64_t o_row_stride, int64_t o_head_stride, int64_t lse_stride,
                       void* stream);
int lvattn_combine_sum(const float* ws, void* out, const int32_t* comb_dev, int n_comb, int block_rows, int heads, int d,
                       int64_t o_row_stride, int64_t o_head_stride, void* stream);

/* ---- incremental attention over a KV cache -------------------------------------
 * Replaces torch_npu.npu_incre_flash_attention as called with --use-kv-cache
 * (lcvlm_modellink/core/transformer/dot_product_attention.py:428-438: query [b, 1, hq*d], key / value
 * [b, k_len, hkv*d], layout "BSH", no mask).  q [batch, hq, d], k / v [batch, k_len, hkv, d], out [batch, hq, d]
 * bf16 with element strides (last dimension contiguous); lse [batch, hq] fp32 (natural log) or NULL.
 * The key range is cut into n_split pieces (0 = chosen from the device size), each read once by one
 * workgroup; `workspace` (lvattn_decode_workspace_bytes(...) bytes of device memory, contents undefined
 * afterwards) holds the pieces' partial results until a second small kernel merges them. */
int64_t lvattn_decode_workspace_bytes(int batch, int hq, int hkv, int d);
int lvattn_decode(const void* q, const void* k, const void* v, void* out, float* lse, void* workspace,
                  int64_t workspace_bytes, int batch, int k_len, int hq, int hkv, int d,
                  int64_t q_batch_stride, int64_t q_head_stride,
                  int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                  int64_t v_batch_stride, int64_t v_row_stride, int64_t v_head_stride,
                  int64_t o_batch_stride, int64_t o_head_stride,
                  float softmax_scale, int n_split, void* stream);

/* lvattn_decode without any allocation, by the caller or the library: the key ranges' partial results go to a fixed
 * buffer inside the library image (lvattn_decode_noalloc_max_bytes() = 8 MiB per GPU, held from the first call on
 * that GPU until the process ends) instead of a caller's workspace.  No counterpart in the reference; it is for
 * callers that cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  Same arguments and results as lvattn_decode minus the workspace.  Returns LV_ESIZE when
 * batch * hkv * n_split * (hq / hkv) * (d + 1) * 4 bytes exceed the buffer (pass a smaller n_split or use
 * lvattn_decode).  For 40 / 8 heads at d = 128 that is batch * n_split <= 406; the default split keeps
 * batch * hkv * n_split near 512 (768 for caches of 393,216 tokens and more: n_split = ceil(512 or 768 / (batch * hkv)),
 * at least 1), i.e. at most about 2 MiB of partial results, so the default fits at any batch up to 406 and only an
 * explicit large n_split overflows.
 * The first call on each GPU looks the buffer's device address up (hipGetSymbolAddress, which may load the code
 * object): make that first call OUTSIDE stream capture; later calls on that GPU are capture-safe.  All calls on one
 * GPU share the buffer: calls that overlap on one GPU (two streams, two threads) race and must be serialised by the
 * caller. */
int64_t lvattn_decode_noalloc_max_bytes(void);
int lvattn_decode_noalloc(const void* q, const void* k, const void* v, void* out, float* lse,
                          int batch, int k_len, int hq, int hkv, int d,
                          int64_t q_batch_stride, int64_t q_head_stride,
                          int64_t k_batch_stride, int64_t k_row_stride, int64_t k_head_stride,
                          int64_t v_batch_stride, int64_t v_row_stride, int64_t v_head_stride,
                          int64_t o_batch_stride, int64_t o_head_stride,
                          float softmax_scale, int n_split, void* stream);

/* ---- attention backward -----------------------------------------------------
 * Replaces the autograd of the calls above (npu_fusion_attention_grad, reached
 * through torch autograd from dot_product_attention.py:352-365 / :314-325) and the
 * per-step backward of the ring.
 * Step 1: lvattn_bwd_delta: delta[h, row] = sum_d dout*out (fp32 [hq, lse_stride]).
 * Step 2: lvattn_bwd launches two kernels, neither of which sums across workgroups
 *         (no atomics; results are bitwise reproducible):
 *           dK/dV: one work item per key block    (descriptors from lv_build_bwd_blocks*)
 *           dQ   : one work item per query block  (descriptors from lv_build_fwd_blocks*)
 *         With accum_f32 == 0, dq [rows,hq,d] / dk / dv [rows,hkv,d] are bf16 and written;
 *         with accum_f32 == 1 they are fp32 buffers that are ADDED to (ring steps); with accum_f32 == 2
 *         fp32 buffers that are written (the partial results of split sweeps: no zero fill, no read).
 *         Strides of dq/dk/dv are in elements of their own type. */
int lvattn_bwd_delta(const void* out, const void* dout, float* delta,
                     int rows, int hq, int d,
                     int64_t o_row_stride, int64_t o_head_stride,
                     int64_t do_row_stride, int64_t do_head_stride,
                     int64_t lse_stride, void* stream);
int lvattn_bwd(const void* q, const void* k, const void* v, const void* dout,
               const float* lse, const float* delta,
               void* dq, void* dk, void* dv,
               const int32_t* qdesc_dev, int n_qblocks, const int32_t* kdesc_dev, int n_kblocks,
               int q_rows_total, int k_rows_total, int hq, int hkv, int d,
               int64_t q_row_stride, int64_t q_head_stride,
               int64_t k_row_stride, int64_t k_head_stride,
               int64_t v_row_stride, int64_t v_head_stride,
               int64_t do_row_stride, int64_t do_head_stride,
               int64_t dq_row_stride, int64_t dq_head_stride,
               int64_t dk_row_stride, int64_t dk_head_stride,
               int64_t dv_row_stride, int64_t dv_head_stride,
               int64_t lse_stride, float softmax_scale, int causal, int accum_f32,
               void* stream);
/* fp32 [rows, heads, d] contiguous -> bf16 with strides (casting the ring's fp32 accumulators). */
int lvattn_cast_f32_bf16(const float* src, void* dst, int rows, int heads, int d,
                         int64_t dst_row_stride, int64_t dst_head_stride, void* stream);

/* ---- ring merge -------------------------------------------------------------
 * The (out, softmax_max, softmax_sum) update MindSpeed's ringattn_context_parallel
 * performs after each ring step, in log-sum-exp form:
 *   lse' = log(exp(lse_acc) + exp(lse_blk));  out_acc = out_acc*exp(lse_acc-lse') + out_blk*exp(lse_blk-lse')
 * out_acc fp32 [rows, hq, d] contiguous, lse_acc fp32 [hq, lse_stride], updated in
 * place for rows [row0, row0+rows); out_blk bf16 [.., hq, d] / lse_blk indexed by the
 * same absolute rows. */
int lvattn_merge(float* out_acc, float* lse_acc, const void* out_blk, const float* lse_blk,
                 int row0, int rows, int hq, int d,
                 int64_t blk_row_stride, int64_t blk_head_stride,
                 int64_t lse_stride, void* stream);
/* The first step of a ring pass: (out_acc, lse_acc) := (out_blk, lse_blk) for the same rows, without reading the
 * accumulators (they need no zero / -inf fill beforehand). */
int lvattn_merge_init(float* out_acc, float* lse_acc, const void* out_blk, const float* lse_blk,
                      int row0, int rows, int hq, int d,
                      int64_t blk_row_stride, int64_t blk_head_stride,
                      int64_t lse_stride, void* stream);

/* ---- token compression / indexing --------------------------------------------
 * lv_pixel_shuffle: MegatronVisionModel.forward_once + pixel_shuffle(0.5)
 * (lcvlm_modellink/pretrain_lcvlm.py:463-470, 546-556).  in [n, tokens_in, c] bf16
 * where the first `skip` tokens of every image (the class token) are dropped and
 * the remaining side*side tokens form a square grid; out [n, (side/2)^2, 4c]. */
int lv_pixel_shuffle(const void* in, void* out, int n, int side, int c, int skip,
                     int64_t in_image_stride, void* stream);
/* lv_pixel_shuffle_v2: the other token compression of MegatronVisionModel.forward_once, taken with
 * --vision-downsample-stride (lcvlm_modellink/pretrain_lcvlm.py:472-476, pixel_shuffle_v2 :558-570): every
 * scale_stride x scale_stride patch of the side x side token grid becomes one token of scale_stride^2 * c channels, the
 * patch's tokens in (row, column) order, zeros where a patch hangs over the right / bottom edge.  in as for
 * lv_pixel_shuffle (the first `skip` tokens of an image dropped); out [n, ceil(side/scale_stride)^2, scale_stride^2 * c]. */
int lv_pixel_shuffle_v2(const void* in, void* out, int n, int side, int c, int skip, int scale_stride,
                        int64_t in_image_stride, void* stream);
/* lv_scatter_rows: LanguageModelEmbedding.forward visual-feature scatter
 * (lcvlm_modellink/core/models/common/embeddings/language_model_embedding.py:123,131):
 * dst[dst_index[i]] = src[src_index[i]] for i < n, rows of `width` bf16 elements
 * (src_index may be NULL = identity).  Indices are flat row numbers (b*s + pos). */
int lv_scatter_rows(const void* src, void* dst, const int64_t* src_index, const int64_t* dst_index,
                    int64_t n, int width, int64_t src_rows, int64_t dst_rows, void* stream);
/* lv_gather_rows: zig-zag context-parallel slicing of activations / RoPE tables
 * (lcvlm_modellink/training/utils.py:351-362, rotary_pos_embedding.py:36-47,114-117):
 * dst[i] = src[index[i]], rows of `row_bytes` bytes (multiple of 4). */
int lv_gather_rows(const void* src, void* dst, const int64_t* index, int64_t n,
                   int64_t row_bytes, int64_t src_rows, void* stream);

/* ---- rotary position embedding ---------------------------------------------------
 * lv_rope: apply_rotary_pos_emb_bshd / apply_rotary_pos_emb_thd
 * (lcvlm_modellink/core/models/common/embeddings/rotary_pos_embedding.py:181-204, 207-229; called on
 * query and key by Megatron's SelfAttention ahead of flash_attention_forward), non-interleaved form.
 * x / out [rows, heads, d] bf16 with row and head strides (out may alias x); the first `rot` columns are
 * rotated, the rest copied.  cos_tab / sin_tab [tab_rows, rot] bf16: cos and sin of the angle table
 * RotaryEmbedding.forward builds (:84-122), rounded to bf16 as :200-201 does.  The table row of a tensor
 * row is pos_ids[row] when pos_ids is given (packed sequences: positions restart per sample, :114-117 /
 * the per-segment freqs[:len] of :223-229), else row / tab_div (tab_div = batch size when the table has
 * one row per sequence position of an [s, b, h, d] tensor).  Every product and the sum round to bf16 as
 * the reference's eager ops do; conj != 0 applies the transposed rotation (the gradient). */
int lv_rope(const void* x, void* out, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids,
            int64_t rows, int heads, int d, int rot,
            int64_t x_row_stride, int64_t x_head_stride, int64_t o_row_stride, int64_t o_head_stride,
            int64_t tab_rows, int tab_div, int conj, void* stream);
/* The rotation folded into the attention path (the scripts' --use-fused-rotary-pos-emb,
 * scripts/modellink/qwen25/finetune_qwen25_14b_intern_300m_ptd_tp8pp1_stage4.sh:164, removes the same pass on the NPU):
 *  - lvattn_fwd_rope = lvattn_fwd on UNROTATED queries: each workgroup rotates its query rows in registers as it
 *    loads them (same tables, table-row rule and bf16 roundings as lv_rope with rot == d, so the result equals
 *    lvattn_fwd on lv_rope(q) bit for bit when that launch runs the ping-pong kernel too; lvattn_fwd's default for key
 *    sweeps of 32,768 and more is the one-wave kernel, which has no rotation and rounds differently, a few 1e-3
 *    relative); k must already be rotated.  q_rot (optional, [q rows, hq, d] bf16 with
 *    strides) receives the rotated queries for the backward pass.  Head dim 128 only.
 *  - lv_rope_pack_kv rotates K while packing K and V into the ring's message buffer kv [2, rows, heads, d]
 *    (contiguous): kv[0] = lv_rope(k), kv[1] = v -- one pass instead of a rotation and two copies. */
int lvattn_fwd_rope(const void* q, const void* k, const void* v, void* out, float* lse,
                    const int32_t* desc_dev, int n_blocks, int q_rows_total, int k_rows_total,
                    int hq, int hkv, int d,
                    int64_t q_row_stride, int64_t q_head_stride,
                    int64_t k_row_stride, int64_t k_head_stride,
                    int64_t v_row_stride, int64_t v_head_stride,
                    int64_t o_row_stride, int64_t o_head_stride,
                    int64_t lse_stride, float softmax_scale, int causal,
                    const void* cos_tab, const void* sin_tab, const int32_t* pos_ids, int64_t tab_rows, int tab_div,
                    void* q_rot, int64_t q_rot_row_stride, int64_t q_rot_head_stride, void* stream);
int lv_rope_pack_kv(const void* k, const void* v, void* kv, const void* cos_tab, const void* sin_tab,
                    const int32_t* pos_ids, int64_t rows, int heads, int d, int rot,
                    int64_t k_row_stride, int64_t k_head_stride, int64_t v_row_stride, int64_t v_head_stride,
                    int64_t tab_rows, int tab_div, void* stream);

/* ---- InternViT encoder pieces around its GEMMs --------------------------------------
 * The reference builds the tower from Megatron modules (lcvlm_modellink/core/models/vision/vit_layer_specs.py:80-103:
 * TENorm / ColumnParallelLinear / MLP with torch.nn.functional.gelu, pretrain_lcvlm.py:191-213) wired by
 * InternViTTransformerLayer.forward (intern_vit_model.py:32-89) and InternViTModel.forward (:190-261).  All bf16,
 * fp32 arithmetic, rows of `h` contiguous elements, 16-byte aligned.
 * lv_layernorm: input_layernorm / pre_mlp_layernorm (:48, :69): y = (x-mean)/sqrt(var+eps)*gamma+beta per row.
 * lv_bias_gelu: the MLP's activation on linear_fc1's output plus its bias (:72): y = gelu_erf(x + bias); bias may be
 *   NULL (the projector's MLP has none, pretrain_lcvlm.py:394).
 * lv_bias_gelu_bwd / lv_layernorm_bwd: the gradients torch autograd derives for those two in the reference (the
 *   projector of pretrain_lcvlm.py:431-451 trains in every stage): dx = dy * gelu'(x + bias); LayerNorm dx plus
 *   dgamma_dbeta [2, h] fp32 (dgamma then dbeta), summed over rows in a fixed order through `workspace`
 *   (lv_layernorm_bwd_workspace_bytes(rows, h) bytes of device memory, contents undefined afterwards).
 * lv_bias_scale_residual: `residual + (out + bias) * ls` (:59-63, :74-77); bias may be NULL.
 * lv_bias_scale_residual_layernorm: the same followed by the NEXT sub-block's LayerNorm (:69 after :59-66; the next
 *   layer's :48 after :74-77) in one pass: out = residual + (x + bias) * ls, normed = LayerNorm(out); bit-identical to
 *   the two calls, one read of `out` less.
 * lv_patchify: the patch side of conv1 (:139-145, :203-205) as a GEMM operand: [n, c, H, W] ->
 *   [n*(H/p)*(W/p), kpad], column (ch*p + ky)*p + kx (= conv1.weight.view(out, -1)), zero-padded to kpad.
 * lv_embed_tokens: class-token concat, conv bias, position embedding and the [b,s,h] -> [s,b,h] permute
 *   (:207-216, :235-237): proj [n, n_patches, h] -> tokens [n_class + n_patches, n, h]. */
int lv_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int h,
                 int64_t x_row_stride, int64_t y_row_stride, float eps, void* stream);
int lv_bias_gelu(const void* x, const void* bias, void* y, int64_t rows, int h, void* stream);
int lv_bias_gelu_bwd(const void* x, const void* bias, const void* dy, void* dx, int64_t rows, int h, void* stream);
int64_t lv_layernorm_bwd_workspace_bytes(int64_t rows, int h);
int lv_layernorm_bwd(const void* x, const void* dy, const void* gamma, void* dx, float* dgamma_dbeta, void* workspace,
                     int64_t workspace_bytes, int64_t rows, int h, int64_t x_row_stride, int64_t dy_row_stride,
                     int64_t dx_row_stride, float eps, void* stream);
int lv_bias_scale_residual(const void* x, const void* bias, const void* ls, const void* residual, void* out,
                           int64_t rows, int h, void* stream);
int lv_bias_scale_residual_layernorm(const void* x, const void* bias, const void* ls, const void* residual, void* out,
                                     void* normed, const void* gamma, const void* beta, int64_t rows, int h, float eps,
                                     void* stream);
int lv_patchify(const void* img, void* patches, int n, int c, int height, int width, int patch, int kpad, void* stream);
int lv_embed_tokens(const void* proj, const void* conv_bias, const void* class_token, const void* pos_emb, void* tokens,
                    int n, int n_patches, int n_class, int h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LVATTN_H */
