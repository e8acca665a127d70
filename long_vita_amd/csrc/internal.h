// Internal launch interfaces between the translation units of liblvattn.
#pragma once
#include "common.h"

struct FwdParams {
    const __bf16* q;
    const __bf16* k;
    const __bf16* v;
    __bf16* o;
    float* lse;
    const AttnDesc* desc;
    int n_blocks, hq, hkv, group;  // group = hq / hkv
    int q_rows_total, k_rows_total;
    int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, o_rs, o_hs, lse_stride;
    float scale, scale_log2e;
    int head_per_xcd;   // workgroup order: every work item of a KV head on ONE XCD, back to back (set by the launchers)
    // rotary embedding applied to the query fragments as they are loaded (null = off): bf16 tables [tab_rows][d],
    // table row of tensor row i = pos[i] or i / tab_div; q_rot (optional) receives the rotated queries
    const __bf16* rope_cos;
    const __bf16* rope_sin;
    const int32_t* rope_pos;
    int64_t rope_tab_rows;
    int rope_tab_div;
    __bf16* q_rot;
    int64_t qr_rs, qr_hs;
};
// Workgroup i of a launch runs on XCD i % 8, and every XCD has its own L2.  When ALL the work items of one KV head
// (query blocks x the group's query heads) fit on an XCD's 32 CUs at once, they are given consecutive slots of one XCD:
// the head's K / V then come into that L2 once instead of once per XCD that hosts one of its items (the vision
// tower: 1024 heads x 4-5 query blocks, where the plain order fetched K / V four times, L2 hit rate 41 %).
inline bool lv_head_per_xcd(const FwdParams& p) { return (int64_t)p.n_blocks * p.group <= 32 && p.hkv >= 8; }
#ifdef __HIPCC__
// Work item `blk` of a forward launch: the caller's descriptor, or (desc == nullptr: ONE dense segment, no descriptor
// memory at all) rows [blk * bm, + bm) of q_rows_total query rows against all k_rows_total keys; the causal rule is
// aligned to the END of both ranges (query row i sees keys j <= i + k_rows_total - q_rows_total).  Causal blocks are
// numbered from the last one down: heaviest first, as the host block builders order theirs.
__device__ __forceinline__ AttnDesc lv_fwd_desc(const FwdParams& p, int blk, int bm, bool causal) {
    if (p.desc) return p.desc[blk];
    if (causal) blk = p.n_blocks - 1 - blk;
    AttnDesc d;
    d.a_row0 = blk * bm;
    d.a_rows = min(bm, p.q_rows_total - d.a_row0);
    d.b_row0 = 0;
    d.out_delta = 0;
    d.band_p1 = 0;
    d.shift = causal ? p.k_rows_total - p.q_rows_total + d.a_row0 : 0;
    d.b_len = causal ? max(0, min(p.k_rows_total, d.a_rows + d.shift)) : p.k_rows_total;
    return d;
}
#endif
int lv_launch_fwd(const FwdParams& p, int d, int causal, int waves, hipStream_t stream);
int lv_launch_fwd_pp(const FwdParams& p, int d, int causal, hipStream_t stream);  // ping-pong form, 256-row items
int lv_launch_fwd_pp16(const FwdParams& p, int causal, hipStream_t stream);      // ping-pong form on 16x16x32 MFMAs, d = 128
int lv_launch_fwd_a4(const FwdParams& p, int causal, hipStream_t stream);        // one wave per SIMD, 64 query rows per wave, d = 128

// Incremental attention over a KV cache: q [batch, hq, d], k/v [batch, k_len, hkv, d], out [batch, hq, d].
#define LV_DECODE_MAX_SPLITS 256
#define LV_DECODE_TILE 128
#define LV_DECODE_SCRATCH_BYTES (8 << 20)   // partial results of lvattn_decode_noalloc, held by every GPU the library runs on
#define LV_MAX_DEVICES 64
struct DecodeParams {
    const __bf16* q;
    const __bf16* k;
    const __bf16* v;
    __bf16* o;
    float* lse;        // [batch, hq] or null
    float* part_o;     // [batch*hkv][n_split][group][d]
    float* part_lse;   // [batch*hkv][n_split][group]
    int batch, hq, hkv, group, k_len, chunk, n_split;
    int64_t q_bs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_hs;
    float scale, scale_log2e;
};
int lv_launch_decode(const DecodeParams& p, int d, hipStream_t stream);

#define LV_BWD_BLOCK_N_INTERNAL 128

struct BwdParams {
    const __bf16* q;
    const __bf16* k;
    const __bf16* v;
    const __bf16* dout;
    const float* lse;
    const float* delta;
    void* dq;    // bf16 (stored) or fp32 (accumulated), as dk / dv
    void* dk;
    void* dv;
    const AttnDesc* desc;    // key blocks (dK/dV kernel)
    const AttnDesc* qdesc;   // query blocks (dQ kernel)
    int n_blocks, n_qblocks, hq, hkv, group;
    int q_rows_total, k_rows_total;
    int64_t q_rs, q_hs, k_rs, k_hs, v_rs, v_hs, do_rs, do_hs, dq_rs, dq_hs, dk_rs, dk_hs, dv_rs, dv_hs, lse_stride;
    float scale, scale_log2e;
    int f32_store;   // fp32 results are stored, not added to what the buffers hold (accum_f32 == 2)
};
int lv_launch_bwd(const BwdParams& p, int d, int causal, int accum, hipStream_t stream);
int lv_launch_bwd_dkdv_rs(const BwdParams& p, int d, int causal, int accum, hipStream_t stream);  // role-split dK/dV kernel
int lv_launch_bwd_dkdv_g(const BwdParams& p, int causal, int accum, hipStream_t stream);          // one wave per SIMD, gap-scheduled dK/dV kernel, d = 128
int lv_launch_bwd_dq_pp(const BwdParams& p, int d, int causal, int accum, hipStream_t stream);  // ping-pong dQ kernel
void lv_select_bwd_kernels(int dq, int dkdv);   // 0 default, 1 the earlier kernel, (dkdv) 2 the gap-scheduled kernel, < 0 unchanged
void lv_select_rs_units(int n);                  // 32-row units per iteration of the role-split dK/dV kernel: 1 or 2
int lv_launch_bwd_delta(const void* o, const void* dout, float* delta, int rows, int hq, int d, int64_t o_rs, int64_t o_hs,
                        int64_t do_rs, int64_t do_hs, int64_t lse_stride, hipStream_t stream);
int lv_launch_rope(const void* x, void* out, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids, int64_t rows,
                   int heads, int d, int rot, int64_t x_rs, int64_t x_hs, int64_t o_rs, int64_t o_hs, int64_t tab_rows, int tab_div,
                   int conj, hipStream_t stream);
int lv_launch_rope_pack_kv(const void* k, const void* v, void* kv, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids,
                           int64_t rows, int heads, int d, int rot, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                           int64_t tab_rows, int tab_div, hipStream_t stream);
int lv_launch_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int h, int64_t x_rs,
                        int64_t y_rs, float eps, hipStream_t stream);
int lv_launch_bias_gelu(const void* x, const void* bias, void* y, int64_t rows, int h, hipStream_t stream);
int lv_launch_bias_gelu_bwd(const void* x, const void* bias, const void* dy, void* dx, int64_t rows, int h, hipStream_t stream);
int lv_layernorm_bwd_parts(int64_t rows);
int lv_launch_layernorm_bwd(const void* x, const void* dy, const void* gamma, void* dx, float* dgamma_dbeta, float* part,
                            int64_t rows, int h, int64_t x_rs, int64_t dy_rs, int64_t dx_rs, float eps, hipStream_t stream);
int lv_launch_bias_scale_residual(const void* x, const void* bias, const void* ls, const void* res, void* out, int64_t rows, int h,
                                  hipStream_t stream);
int lv_launch_bsr_layernorm(const void* x, const void* bias, const void* ls, const void* res, void* x1, void* y, const void* gamma,
                            const void* beta, int64_t rows, int h, float eps, hipStream_t stream);
int lv_launch_patchify(const void* img, void* out, int n, int c, int H, int W, int p, int kpad, hipStream_t stream);
int lv_launch_embed_tokens(const void* proj, const void* conv_bias, const void* cls, const void* pos, void* out, int n, int np,
                           int ncls, int h, hipStream_t stream);
