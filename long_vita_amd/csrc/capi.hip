// extern "C" entry points of liblvattn (declared in include/lvattn.h), the host-side
// block builders, and the small bandwidth-bound kernels (ring merge, pixel shuffle,
// row scatter / gather, fp32 -> bf16 cast).
#include "../../include/lvattn.h"
#include "internal.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int lvattn_abi_version(void) { return 1; }

// Forward kernel choice.  Default: the one-wave-per-SIMD kernel ("a4") for head dim 128 when a work item sweeps at least
// 32,768 keys on average (shorter sweeps take "pp", see fwd_a4 below) and the lockstep kernel for head dim 64; the ping-pong
// kernel ("pp") carries the fused rotary embedding and stays selectable, as do pp16 and
// lockstep (a4 +1.8 % over pp at 1M tokens, pp +13 % over lockstep at d = 128, lockstep +16 % over pp at d = 64).  The
// environment variable LVATTN_FWD_KERNEL=a4 / pp / pp16 / lockstep forces one (a4 and pp16 exist for d = 128 only:
// other head dims then take their default).
static int fwd_kernel_code(const char* e) {   // 0 default, 1 ping-pong (32x32x16 MFMAs), 2 lockstep, 3 ping-pong on 16x16x32 MFMAs, 4 one wave per SIMD
    if (!e) return 0;
    return !strcmp(e, "pp") ? 1 : (!strcmp(e, "lockstep") ? 2 : (!strcmp(e, "pp16") ? 3 : (!strcmp(e, "a4") ? 4 : 0)));
}
static int g_fwd_kernel = -1;   // -1: not read yet
static int fwd_kernel_override() {
    if (g_fwd_kernel < 0) g_fwd_kernel = fwd_kernel_code(getenv("LVATTN_FWD_KERNEL"));
    return g_fwd_kernel;
}
extern "C" int lvattn_select_fwd_kernel(const char* name) {
    if (name && name[0] && strcmp(name, "default") && !fwd_kernel_code(name)) return fail(LV_EINVAL, "unknown forward kernel '%s' (default, pp, pp16, a4, lockstep)", name);
    g_fwd_kernel = fwd_kernel_code(name);
    return LV_OK;
}
extern "C" int lvattn_select_bwd_kernels(const char* dq, const char* dkdv) {
    int a = 0, b = 0, units = 2;
    if (dq && dq[0] && strcmp(dq, "default") && strcmp(dq, "pp")) {
        if (strcmp(dq, "lockstep")) return fail(LV_EINVAL, "unknown dQ kernel '%s' (default, pp, lockstep)", dq);
        a = 1;
    }
    if (dkdv && dkdv[0] && strcmp(dkdv, "default")) {
        if (!strcmp(dkdv, "single")) b = 1;
        else if (!strcmp(dkdv, "g")) b = 2;
        else if (!strcmp(dkdv, "rs")) b = 3;
        else if (!strcmp(dkdv, "rs1")) b = 3, units = 1;
        else return fail(LV_EINVAL, "unknown dK/dV kernel '%s' (default, rs, rs1, single, g)", dkdv);
    }
    lv_select_bwd_kernels(a, b);
    lv_select_rs_units(units);
    return LV_OK;
}
static bool fwd_pp16(int d) { return d == 128 && fwd_kernel_override() == 3; }
// (the one-wave kernel stages K / V rows at 256-byte granularity: row strides that are not multiples of 128 elements, views
// no real layout of this path produces, take the ping-pong kernel.  When the choice is the library's, what decides is the key
// sweep of a work item: one wave per SIMD leaves a workgroup's prologue uncovered, so short sweeps take the ping-pong kernel.
// Same-box A/B at 131,072 packed tokens, equal segments: 1K-token segments a4 21 % slower, 8K 8 %, 32K 0.9 %, 64K (mean sweep
// 32K keys) a4 0.6 % faster; single segments the same: 8K rows 7 % slower, 32K 0.7 %, 128K 0.5 % faster.  `mean_keys` is the
// caller's figure (lv_mean_keys_per_item over the descriptors it built); without one a causal launch is taken for ONE
// segment (mean sweep = half the rows), which is what a packed launch through plain lvattn_fwd is then mistaken for)
static const int kA4MinSweep = 32768;
static bool fwd_a4(int d, int64_t k_rs, int64_t v_rs, int q_rows_total, int k_rows_total, int causal, int mean_keys) {
    const int o = fwd_kernel_override();
    if (d != 128 || k_rs % 128 || v_rs % 128) return false;
    if (o == 4) return true;
    if (o != 0) return false;
    if (mean_keys >= 0) return mean_keys >= kA4MinSweep;
    return causal ? q_rows_total >= 2 * kA4MinSweep : k_rows_total >= kA4MinSweep;
}
static bool fwd_pingpong(int d) { return fwd_kernel_override() == 1 || (d == 128 && (fwd_kernel_override() == 4 || fwd_kernel_override() == 0)); }
// Query rows per forward work item: 256 (8 waves per workgroup); with the lockstep kernel forced,
// LVATTN_FWD_WAVES=4 selects 128-row items (4 waves, two workgroups per CU).
// Follows lvattn_select_fwd_kernel: descriptors must be (re)built with the block size in force at launch time.
static int fwd_waves() {
    static const bool four = [] {
        const char* e = getenv("LVATTN_FWD_WAVES");
        return e && atoi(e) == 4;
    }();
    return (four && fwd_kernel_override() == 2) ? 4 : 8;
}
extern "C" int lvattn_fwd_block_m(void) { return fwd_waves() * 32; }
extern "C" const char* lvattn_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------
// host block builders
// ---------------------------------------------------------------------------
namespace {
struct HostDesc {
    int32_t v[LV_DESC_INTS];
    int64_t work;
};

int emit(std::vector<HostDesc>& blocks, int32_t* out, int max_blocks) {
    // heaviest first: the tail of the launch is then made of the cheapest work items
    std::stable_sort(blocks.begin(), blocks.end(), [](const HostDesc& a, const HostDesc& b) { return a.work > b.work; });
    if ((int)blocks.size() > max_blocks) return fail(LV_ESIZE, "need %d block descriptors, room for %d", (int)blocks.size(), max_blocks);
    for (size_t i = 0; i < blocks.size(); ++i) memcpy(out + i * LV_DESC_INTS, blocks[i].v, sizeof(int32_t) * LV_DESC_INTS);
    return (int)blocks.size();
}

// forward: query blocks of a rectangle q[q0, q0+ql) x k[k0, k0+kl)
// (band >= 0: a row also sees no key more than `band` positions before its last visible one; the sweep of a block then
// starts at the first 64-key tile its first row needs)
void fwd_rect(std::vector<HostDesc>& blocks, int q0, int ql, int k0, int kl, int causal, int shift, int band = -1) {
    const int BM = lvattn_fwd_block_m();
    for (int m = 0; m < ql; m += BM) {
        HostDesc d{};
        int rows = std::min(BM, ql - m);
        int klen = kl;
        int sh = 0;
        if (causal) {
            sh = shift + m;  // key j visible to local row i iff j <= i + m + shift
            int64_t vis = (int64_t)rows + sh;  // keys visible to the last row
            klen = (int)std::max<int64_t>(0, std::min<int64_t>(kl, vis));
        }
        int kstart = 0;
        if (causal && band >= 0) {
            kstart = (int)std::min<int64_t>(std::max<int64_t>(0, (int64_t)sh - band), klen) & ~63;
            d.v[6] = band + 1;
        }
        d.v[0] = q0 + m; d.v[1] = rows; d.v[2] = k0 + kstart; d.v[3] = klen - kstart; d.v[4] = sh - kstart;
        d.work = (int64_t)rows * (klen - kstart);
        blocks.push_back(d);
    }
}

// backward: key blocks of the same rectangle; query rows that see at least one key of
// the block start at i >= j0 - shift.
void bwd_rect(std::vector<HostDesc>& blocks, int q0, int ql, int k0, int kl, int causal, int shift, int band = -1) {
    for (int n = 0; n < kl; n += LV_BWD_BLOCK_N) {
        HostDesc d{};
        int rows = std::min(LV_BWD_BLOCK_N, kl - n);
        int qstart = 0;
        int sh = 0x3fffffff;
        if (causal) {
            // key j = n + jj visible to query i iff n + jj <= i + shift
            qstart = std::max(0, n - shift);
            qstart = std::min(qstart, ql);
            qstart &= ~31;                 // keep 32-row slices aligned to the block start
            sh = shift + qstart - n;       // jj <= ii + sh with ii relative to qstart
        }
        int qlen = ql - qstart;
        if (causal && band >= 0) {
            // query ii (relative to qstart) sees key jj of the block only while jj >= ii + sh - band: the last such row is
            // ii = rows - 1 - sh + band
            qlen = (int)std::max<int64_t>(0, std::min<int64_t>(qlen, (int64_t)rows - sh + band));
            d.v[6] = band + 1;
        }
        d.v[0] = k0 + n; d.v[1] = rows; d.v[2] = q0 + qstart; d.v[3] = qlen; d.v[4] = sh;
        d.work = (int64_t)rows * qlen;
        blocks.push_back(d);
    }
}

int check_cu(const int32_t* cu, int n_seg) {
    if (!cu || n_seg < 0) return fail(LV_EINVAL, "cu_seqlens is null or n_seg < 0");
    if (n_seg > 0 && cu[0] < 0) return fail(LV_EINVAL, "cu_seqlens[0] < 0");
    for (int i = 0; i < n_seg; ++i)
        if (cu[i + 1] < cu[i]) return fail(LV_EINVAL, "cu_seqlens not non-decreasing at %d", i);
    return 0;
}
}  // namespace

extern "C" int lv_build_fwd_blocks(const int32_t* cu, int n_seg, int causal, int32_t* out, int max_blocks) {
    if (int e = check_cu(cu, n_seg)) return e;
    std::vector<HostDesc> blocks;
    for (int i = 0; i < n_seg; ++i) fwd_rect(blocks, cu[i], cu[i + 1] - cu[i], cu[i], cu[i + 1] - cu[i], causal, 0);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_bwd_blocks(const int32_t* cu, int n_seg, int causal, int32_t* out, int max_blocks) {
    if (int e = check_cu(cu, n_seg)) return e;
    std::vector<HostDesc> blocks;
    for (int i = 0; i < n_seg; ++i) bwd_rect(blocks, cu[i], cu[i + 1] - cu[i], cu[i], cu[i + 1] - cu[i], causal, 0);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_fwd_blocks_band(const int32_t* cu, int n_seg, int band, int32_t* out, int max_blocks) {
    if (int e = check_cu(cu, n_seg)) return e;
    if (band < 0) return fail(LV_EINVAL, "band %d < 0", band);
    std::vector<HostDesc> blocks;
    for (int i = 0; i < n_seg; ++i) fwd_rect(blocks, cu[i], cu[i + 1] - cu[i], cu[i], cu[i + 1] - cu[i], 1, 0, band);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_bwd_blocks_band(const int32_t* cu, int n_seg, int band, int32_t* out, int max_blocks) {
    if (int e = check_cu(cu, n_seg)) return e;
    if (band < 0) return fail(LV_EINVAL, "band %d < 0", band);
    std::vector<HostDesc> blocks;
    for (int i = 0; i < n_seg; ++i) bwd_rect(blocks, cu[i], cu[i + 1] - cu[i], cu[i], cu[i + 1] - cu[i], 1, 0, band);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_fwd_blocks_rect_band(int q0, int ql, int k0, int kl, int shift, int band, int32_t* out, int max_blocks) {
    if (q0 < 0 || ql < 0 || k0 < 0 || kl < 0 || band < 0) return fail(LV_EINVAL, "negative rectangle or band");
    std::vector<HostDesc> blocks;
    fwd_rect(blocks, q0, ql, k0, kl, 1, shift, band);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_fwd_blocks_rect(int q0, int ql, int k0, int kl, int causal, int shift, int32_t* out, int max_blocks) {
    if (q0 < 0 || ql < 0 || k0 < 0 || kl < 0) return fail(LV_EINVAL, "negative rectangle");
    std::vector<HostDesc> blocks;
    fwd_rect(blocks, q0, ql, k0, kl, causal, shift);
    return emit(blocks, out, max_blocks);
}
extern "C" int lv_build_bwd_blocks_rect(int q0, int ql, int k0, int kl, int causal, int shift, int32_t* out, int max_blocks) {
    if (q0 < 0 || ql < 0 || k0 < 0 || kl < 0) return fail(LV_EINVAL, "negative rectangle");
    std::vector<HostDesc> blocks;
    bwd_rect(blocks, q0, ql, k0, kl, causal, shift);
    return emit(blocks, out, max_blocks);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
static int check_common(const void* q, const void* k, const void* v, int hq, int hkv, int d,
                        int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs) {
    if (!q || !k || !v) return fail(LV_EINVAL, "null q/k/v pointer");
    if (d != 64 && d != 128) return fail(LV_EINVAL, "head dim %d not supported (64 or 128)", d);
    if (hq <= 0 || hkv <= 0 || hq % hkv) return fail(LV_EINVAL, "hq=%d must be a positive multiple of hkv=%d", hq, hkv);
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return fail(LV_EINVAL, "q/k/v must be 16-byte aligned");
    if ((q_rs | q_hs | k_rs | k_hs | v_rs | v_hs) & 7) return fail(LV_EINVAL, "row/head strides must be multiples of 8 elements");
    // the kernels address the rows of one tile (at most 256) with 32-bit byte offsets from the tile's first row
    const int64_t rs_max = std::max(std::max(q_rs, k_rs), v_rs), rs_min = std::min(std::min(q_rs, k_rs), v_rs);
    if (rs_min < 0 || rs_max * 2 * 256 > 0x7fffffffLL) return fail(LV_EINVAL, "row stride %lld elements out of range", (long long)(rs_min < 0 ? rs_min : rs_max));
    return 0;
}

struct RopeArgs {
    const void* cos_tab = nullptr;
    const void* sin_tab = nullptr;
    const int32_t* pos_ids = nullptr;
    int64_t tab_rows = 0;
    int tab_div = 1;
    void* q_rot = nullptr;
    int64_t qr_rs = 0, qr_hs = 0;
};

static int fwd_impl(const void* q, const void* k, const void* v, void* out, float* lse,
                    const int32_t* desc_dev, int n_blocks, int q_rows_total, int k_rows_total,
                    int hq, int hkv, int d,
                    int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                    int64_t o_rs, int64_t o_hs, int64_t lse_stride, float scale, int causal, void* stream, const RopeArgs* rope,
                    bool dense = false, int mean_keys = -1) {
    if (int e = check_common(q, k, v, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs)) return e;
    if (!out || !lse || (!desc_dev && !dense)) return fail(LV_EINVAL, "null out/lse/desc pointer");
    if (((uintptr_t)out & 7) || (o_rs & 3) || (o_hs & 3)) return fail(LV_EINVAL, "out must be 8-byte aligned with strides % 4 == 0");
    if (n_blocks < 0 || q_rows_total <= 0 || k_rows_total <= 0) return fail(LV_EINVAL, "bad sizes");
    if (lse_stride < q_rows_total) return fail(LV_EINVAL, "lse_stride %lld < q_rows_total %d", (long long)lse_stride, q_rows_total);
    // One dense segment: the kernels number their work items themselves (lv_fwd_desc), so the launch needs no
    // descriptor memory and its workgroups can be sized to the row count.  A workgroup holds its wave slots for a
    // whole key sweep however few rows it has, so the lockstep kernel runs in 4-wave workgroups (128 rows) when that
    // needs fewer wave slots than 8-wave ones: the vision tower's 1025 rows are 9 x 4 instead of 5 x 8 with seven
    // waves of the last workgroup idle (3.5 % faster on one box; 3-wave workgroups, 11 x 3 with no idle wave at all,
    // measured 7 % slower: more staging per row).
    int dense_waves = 8;
    if (dense) {
        if (causal && k_rows_total < q_rows_total) return fail(LV_EINVAL, "lvattn_fwd_dense: causal needs k_rows >= q_rows (%d < %d)", k_rows_total, q_rows_total);
        // one dense segment: the sweep is known (causal: the rows see k_rows - q_rows + 1 ... k_rows keys)
        mean_keys = causal ? k_rows_total - q_rows_total / 2 : k_rows_total;
        const bool lockstep = !fwd_pp16(d) && !fwd_pingpong(d) && !fwd_a4(d, k_rs, v_rs, q_rows_total, k_rows_total, causal, mean_keys);
        if (lockstep) {
            auto slots = [&](int nw) { return (q_rows_total + 32 * nw - 1) / (32 * nw) * nw; };
            if (fwd_waves() == 4 || slots(4) < slots(8)) dense_waves = 4;
        }
        n_blocks = (q_rows_total + dense_waves * 32 - 1) / (dense_waves * 32);
        desc_dev = nullptr;
    }
    if (n_blocks == 0) return LV_OK;
    if ((int64_t)n_blocks * hq > 0x7fffffffLL) return fail(LV_EINVAL, "grid too large");
    FwdParams p{};
    p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.o = (__bf16*)out; p.lse = lse;
    p.desc = (const AttnDesc*)desc_dev;
    p.n_blocks = n_blocks; p.hq = hq; p.hkv = hkv; p.group = hq / hkv;
    p.q_rows_total = q_rows_total; p.k_rows_total = k_rows_total;
    p.q_rs = q_rs; p.q_hs = q_hs; p.k_rs = k_rs; p.k_hs = k_hs; p.v_rs = v_rs; p.v_hs = v_hs;
    p.o_rs = o_rs; p.o_hs = o_hs; p.lse_stride = lse_stride;
    p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    if (rope) {
        // the rotation lives in the ping-pong kernel (head dim 128), whatever kernel is selected for plain launches
        if (d != 128) return fail(LV_EINVAL, "lvattn_fwd_rope: head dim %d (the fused rotation needs 128)", d);
        if (fwd_waves() != 8) return fail(LV_EINVAL, "lvattn_fwd_rope: needs 256-row work items (LVATTN_FWD_WAVES=4 is set)");
        if (!rope->cos_tab || !rope->sin_tab) return fail(LV_EINVAL, "lvattn_fwd_rope: null cos / sin table");
        if (((uintptr_t)rope->cos_tab | (uintptr_t)rope->sin_tab | (uintptr_t)rope->q_rot) & 15)
            return fail(LV_EINVAL, "lvattn_fwd_rope: tables and q_rot must be 16-byte aligned");
        if (rope->tab_rows <= 0 || rope->tab_div <= 0) return fail(LV_EINVAL, "lvattn_fwd_rope: tab_rows=%lld tab_div=%d", (long long)rope->tab_rows, rope->tab_div);
        if (!rope->pos_ids && (int64_t)(q_rows_total - 1) / rope->tab_div >= rope->tab_rows)
            return fail(LV_EINVAL, "lvattn_fwd_rope: %d rows / %d need more than the table's %lld rows", q_rows_total, rope->tab_div, (long long)rope->tab_rows);
        if (rope->q_rot && ((rope->qr_rs | rope->qr_hs) & 7)) return fail(LV_EINVAL, "lvattn_fwd_rope: q_rot strides must be multiples of 8 elements");
        p.rope_cos = (const __bf16*)rope->cos_tab; p.rope_sin = (const __bf16*)rope->sin_tab; p.rope_pos = rope->pos_ids;
        p.rope_tab_rows = rope->tab_rows; p.rope_tab_div = rope->tab_div;
        p.q_rot = (__bf16*)rope->q_rot; p.qr_rs = rope->qr_rs; p.qr_hs = rope->qr_hs;
        if (lv_launch_fwd_pp(p, d, causal != 0, (hipStream_t)stream)) return fail(LV_ELAUNCH, "forward launch failed: %s", hipGetErrorString(hipGetLastError()));
        return LV_OK;
    }
    if (causal == 2) {     // descriptors with a band: the ping-pong kernel is the one that reads it
        if (dense || fwd_waves() != 8) return fail(LV_EINVAL, "banded launch: needs descriptors of 256-row work items");
        if (lv_launch_fwd_pp(p, d, 1, (hipStream_t)stream)) return fail(LV_ELAUNCH, "forward launch failed: %s", hipGetErrorString(hipGetLastError()));
        return LV_OK;
    }
    const int rc = fwd_a4(d, k_rs, v_rs, q_rows_total, k_rows_total, causal, mean_keys) ? lv_launch_fwd_a4(p, causal, (hipStream_t)stream)
                   : fwd_pp16(d) ? lv_launch_fwd_pp16(p, causal, (hipStream_t)stream)
                   : fwd_pingpong(d) ? lv_launch_fwd_pp(p, d, causal, (hipStream_t)stream)
                                     : lv_launch_fwd(p, d, causal, dense ? dense_waves : fwd_waves(), (hipStream_t)stream);
    if (rc) return fail(LV_ELAUNCH, "forward launch failed: %s", hipGetErrorString(hipGetLastError()));
    return LV_OK;
}

extern "C" int lvattn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                          const int32_t* desc_dev, int n_blocks, int q_rows_total, int k_rows_total,
                          int hq, int hkv, int d,
                          int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                          int64_t o_rs, int64_t o_hs, int64_t lse_stride, float scale, int causal, void* stream) {
    return fwd_impl(q, k, v, out, lse, desc_dev, n_blocks, q_rows_total, k_rows_total, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs,
                    o_rs, o_hs, lse_stride, scale, causal, stream, nullptr);
}

extern "C" int lvattn_fwd_dense(const void* q, const void* k, const void* v, void* out, float* lse,
                                int q_rows, int k_rows, int hq, int hkv, int d,
                                int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                                int64_t o_rs, int64_t o_hs, int64_t lse_stride, float scale, int causal, void* stream) {
    return fwd_impl(q, k, v, out, lse, nullptr, 0, q_rows, k_rows, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs,
                    o_rs, o_hs, lse_stride, scale, causal, stream, nullptr, true);
}

extern "C" int lvattn_fwd_hinted(const void* q, const void* k, const void* v, void* out, float* lse,
                                 const int32_t* desc_dev, int n_blocks, int q_rows_total, int k_rows_total,
                                 int hq, int hkv, int d,
                                 int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                                 int64_t o_rs, int64_t o_hs, int64_t lse_stride, float scale, int causal, int mean_keys_per_item,
                                 void* stream) {
    return fwd_impl(q, k, v, out, lse, desc_dev, n_blocks, q_rows_total, k_rows_total, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs,
                    o_rs, o_hs, lse_stride, scale, causal, stream, nullptr, false, mean_keys_per_item < 0 ? -1 : mean_keys_per_item);
}

extern "C" int lv_mean_keys_per_item(const int32_t* desc_host, int n_blocks) {
    if (!desc_host || n_blocks <= 0) return -1;
    int64_t keys = 0;
    for (int i = 0; i < n_blocks; ++i) keys += std::max<int32_t>(0, desc_host[(size_t)i * LV_DESC_INTS + 3]);
    return (int)(keys / n_blocks);
}

extern "C" int lvattn_fwd_rope(const void* q, const void* k, const void* v, void* out, float* lse,
                               const int32_t* desc_dev, int n_blocks, int q_rows_total, int k_rows_total,
                               int hq, int hkv, int d,
                               int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                               int64_t o_rs, int64_t o_hs, int64_t lse_stride, float scale, int causal,
                               const void* cos_tab, const void* sin_tab, const int32_t* pos_ids, int64_t tab_rows, int tab_div,
                               void* q_rot, int64_t qr_rs, int64_t qr_hs, void* stream) {
    RopeArgs r;
    r.cos_tab = cos_tab; r.sin_tab = sin_tab; r.pos_ids = pos_ids; r.tab_rows = tab_rows; r.tab_div = tab_div;
    r.q_rot = q_rot; r.qr_rs = qr_rs; r.qr_hs = qr_hs;
    return fwd_impl(q, k, v, out, lse, desc_dev, n_blocks, q_rows_total, k_rows_total, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs,
                    o_rs, o_hs, lse_stride, scale, causal, stream, &r);
}

// ---------------------------------------------------------------------------
// incremental attention over a KV cache
// ---------------------------------------------------------------------------
extern "C" int64_t lvattn_decode_workspace_bytes(int batch, int hq, int hkv, int d) {
    if (batch <= 0 || hq <= 0 || hkv <= 0 || d <= 0 || hq % hkv) return 0;
    const int64_t group = hq / hkv;
    return (int64_t)batch * hkv * LV_DECODE_MAX_SPLITS * group * (d + 1) * (int64_t)sizeof(float);
}

// The same kernels with the key ranges' partial results in a buffer that is part of the library image instead of a
// caller's workspace: nothing is allocated, by the caller or here.  Every GPU the library runs on holds its own copy
// of the array at its own address; the array's name is only a host-side handle, the device address comes from
// hipGetSymbolAddress, once per device, with that device current.
static __device__ __align__(256) unsigned char scratch[LV_DECODE_SCRATCH_BYTES];
static std::mutex g_scratch_mutex;
static void* g_scratch_ptr[LV_MAX_DEVICES] = {};

static int scratch_of_current_device(void** ptr) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(LV_ELAUNCH, "lvattn_decode_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= LV_MAX_DEVICES) return fail(LV_EINVAL, "lvattn_decode_noalloc: device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(LV_ELAUNCH, "lvattn_decode_noalloc: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *ptr = g_scratch_ptr[dev];
    return LV_OK;
}

static int decode_impl(const char* who, const void* q, const void* k, const void* v, void* out, float* lse, void* workspace,
                       int64_t workspace_bytes, int batch, int k_len, int hq, int hkv, int d,
                       int64_t q_bs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                       int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_hs,
                       float scale, int n_split, void* stream, bool library_scratch) {
    // q is [batch, hq, d]: its batch stride is no tile row stride (the kernel forms q's offsets in 64 bits), so it only
    // has to be a multiple of 8 elements; the 256-row tile cap of the prefill kernels applies to the K / V row strides
    if (q_bs < 0 || (q_bs & 7)) return fail(LV_EINVAL, "%s: q batch stride %lld must be a non-negative multiple of 8 elements", who, (long long)q_bs);
    if (int e = check_common(q, k, v, hq, hkv, d, 0, q_hs, k_rs, k_hs, v_rs, v_hs)) return e;
    if (!out || (!workspace && !library_scratch)) return fail(LV_EINVAL, "%s: null out/workspace pointer", who);
    if (batch <= 0 || k_len <= 0) return fail(LV_EINVAL, "%s: batch=%d k_len=%d", who, batch, k_len);
    if (hq / hkv > 32) return fail(LV_EINVAL, "%s: %d query heads per KV head (at most 32)", who, hq / hkv);
    if ((k_bs | v_bs) & 7) return fail(LV_EINVAL, "%s: batch strides must be multiples of 8 elements", who);
    if ((uintptr_t)workspace & 15) return fail(LV_EINVAL, "%s: workspace must be 16-byte aligned", who);
    // the tile loads address a split's rows with 32-bit byte offsets
    if ((int64_t)LV_DECODE_TILE * (k_rs > v_rs ? k_rs : v_rs) * 2 > 0x7fffffffLL) return fail(LV_EINVAL, "%s: row stride too large", who);
    const int max_split = (k_len + LV_DECODE_TILE - 1) / LV_DECODE_TILE;
    // Key ranges by default: two workgroups per CU fit at a time, so 512 work items are one round; long caches do better with
    // three per CU (a round and a half: the ranges that finish early leave their CUs to the rest; 1M-token cache 764 -> 722 us,
    // 768K 575 -> 543, 512K 386 -> 374, no gain at 256K and below), and only with whole multiples of the CU count.
    if (n_split <= 0) n_split = ((k_len >= 393216 ? 768 : 512) + batch * hkv - 1) / (batch * hkv);
    if (n_split > max_split) n_split = max_split;
    if (n_split > LV_DECODE_MAX_SPLITS) n_split = LV_DECODE_MAX_SPLITS;
    int chunk = (k_len + n_split - 1) / n_split;
    chunk = (chunk + LV_DECODE_TILE - 1) / LV_DECODE_TILE * LV_DECODE_TILE;
    n_split = (k_len + chunk - 1) / chunk;
    const int group = hq / hkv;
    const int64_t need = (int64_t)batch * hkv * n_split * group * (d + 1) * (int64_t)sizeof(float);
    if (workspace_bytes < need)
        return fail(LV_ESIZE, "%s: %lld bytes of partial results needed (batch %d x %d KV heads x %d key ranges x %d heads x (%d + 1) floats), "
                    "room for %lld", who, (long long)need, batch, hkv, n_split, group, d, (long long)workspace_bytes);
    if ((int64_t)batch * hkv * n_split > 0x7fffffffLL) return fail(LV_EINVAL, "%s: grid too large", who);
    if (library_scratch)   // the first HIP call: every argument has been checked by now
        if (int e = scratch_of_current_device(&workspace)) return e;
    DecodeParams p;
    p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.o = (__bf16*)out; p.lse = lse;
    p.part_o = (float*)workspace;
    p.part_lse = p.part_o + (int64_t)batch * hkv * n_split * group * d;
    p.batch = batch; p.hq = hq; p.hkv = hkv; p.group = group; p.k_len = k_len; p.chunk = chunk; p.n_split = n_split;
    p.q_bs = q_bs; p.q_hs = q_hs; p.k_bs = k_bs; p.k_rs = k_rs; p.k_hs = k_hs; p.v_bs = v_bs; p.v_rs = v_rs; p.v_hs = v_hs;
    p.o_bs = o_bs; p.o_hs = o_hs;
    p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    if (lv_launch_decode(p, d, (hipStream_t)stream)) return fail(LV_ELAUNCH, "%s: launch failed: %s", who, hipGetErrorString(hipGetLastError()));
    return LV_OK;
}

extern "C" int lvattn_decode(const void* q, const void* k, const void* v, void* out, float* lse, void* workspace,
                             int64_t workspace_bytes, int batch, int k_len, int hq, int hkv, int d,
                             int64_t q_bs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                             int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_hs,
                             float scale, int n_split, void* stream) {
    return decode_impl("lvattn_decode", q, k, v, out, lse, workspace, workspace_bytes, batch, k_len, hq, hkv, d, q_bs, q_hs,
                       k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_hs, scale, n_split, stream, false);
}

extern "C" int64_t lvattn_decode_noalloc_max_bytes(void) { return LV_DECODE_SCRATCH_BYTES; }

extern "C" int lvattn_decode_noalloc(const void* q, const void* k, const void* v, void* out, float* lse,
                                     int batch, int k_len, int hq, int hkv, int d,
                                     int64_t q_bs, int64_t q_hs, int64_t k_bs, int64_t k_rs, int64_t k_hs,
                                     int64_t v_bs, int64_t v_rs, int64_t v_hs, int64_t o_bs, int64_t o_hs,
                                     float scale, int n_split, void* stream) {
    // every partial result the merge kernel reads is written by the first kernel of the same call (each key range is
    // non-empty by construction), so nothing of the previous call's contents has to be cleared
    return decode_impl("lvattn_decode_noalloc", q, k, v, out, lse, nullptr, LV_DECODE_SCRATCH_BYTES, batch, k_len, hq, hkv, d,
                       q_bs, q_hs, k_bs, k_rs, k_hs, v_bs, v_rs, v_hs, o_bs, o_hs, scale, n_split, stream, true);
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
extern "C" int lvattn_bwd_delta(const void* out, const void* dout, float* delta, int rows, int hq, int d,
                                int64_t o_rs, int64_t o_hs, int64_t do_rs, int64_t do_hs, int64_t lse_stride, void* stream) {
    if (!out || !dout || !delta) return fail(LV_EINVAL, "null pointer");
    if (d != 64 && d != 128) return fail(LV_EINVAL, "head dim %d not supported (64 or 128)", d);
    if (((uintptr_t)out | (uintptr_t)dout) & 15) return fail(LV_EINVAL, "out/dout must be 16-byte aligned");
    if ((o_rs | o_hs | do_rs | do_hs) & 7) return fail(LV_EINVAL, "strides must be multiples of 8 elements");
    if (rows < 0 || hq <= 0 || lse_stride < rows) return fail(LV_EINVAL, "bad sizes");
    if (rows == 0) return LV_OK;
    if (lv_launch_bwd_delta(out, dout, delta, rows, hq, d, o_rs, o_hs, do_rs, do_hs, lse_stride, (hipStream_t)stream))
        return fail(LV_ELAUNCH, "delta launch failed");
    return LV_OK;
}

extern "C" int lvattn_bwd(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* delta,
                          void* dq, void* dk, void* dv,
                          const int32_t* qdesc_dev, int n_qblocks, const int32_t* kdesc_dev, int n_kblocks,
                          int q_rows_total, int k_rows_total, int hq, int hkv, int d,
                          int64_t q_rs, int64_t q_hs, int64_t k_rs, int64_t k_hs, int64_t v_rs, int64_t v_hs,
                          int64_t do_rs, int64_t do_hs, int64_t dq_rs, int64_t dq_hs,
                          int64_t dk_rs, int64_t dk_hs, int64_t dv_rs, int64_t dv_hs,
                          int64_t lse_stride, float scale, int causal, int accum_f32, void* stream) {
    if (int e = check_common(q, k, v, hq, hkv, d, q_rs, q_hs, k_rs, k_hs, v_rs, v_hs)) return e;
    if (!dout || !lse || !delta || !dq || !dk || !dv) return fail(LV_EINVAL, "null pointer");
    if ((n_qblocks > 0 && !qdesc_dev) || (n_kblocks > 0 && !kdesc_dev)) return fail(LV_EINVAL, "null descriptor pointer");
    if (((uintptr_t)dout & 15) || ((do_rs | do_hs) & 7)) return fail(LV_EINVAL, "dout must be 16-byte aligned, strides % 8 == 0");
    if ((((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) || ((dq_rs | dq_hs | dk_rs | dk_hs | dv_rs | dv_hs) & 3))
        return fail(LV_EINVAL, "dq/dk/dv must be 16-byte aligned with strides % 4 == 0");
    if (n_qblocks < 0 || n_kblocks < 0 || q_rows_total <= 0 || k_rows_total <= 0 || lse_stride < q_rows_total)
        return fail(LV_EINVAL, "bad sizes");
    if (n_qblocks == 0 && n_kblocks == 0) return LV_OK;
    if ((int64_t)n_qblocks * hq > 0x7fffffffLL || (int64_t)n_kblocks * hkv > 0x7fffffffLL) return fail(LV_EINVAL, "grid too large");
    BwdParams p;
    p.q = (const __bf16*)q; p.k = (const __bf16*)k; p.v = (const __bf16*)v; p.dout = (const __bf16*)dout;
    p.lse = lse; p.delta = delta; p.dq = dq; p.dk = dk; p.dv = dv;
    p.desc = (const AttnDesc*)kdesc_dev; p.qdesc = (const AttnDesc*)qdesc_dev;
    p.n_blocks = n_kblocks; p.n_qblocks = n_qblocks; p.hq = hq; p.hkv = hkv; p.group = hq / hkv;
    p.q_rows_total = q_rows_total; p.k_rows_total = k_rows_total;
    p.q_rs = q_rs; p.q_hs = q_hs; p.k_rs = k_rs; p.k_hs = k_hs; p.v_rs = v_rs; p.v_hs = v_hs;
    p.do_rs = do_rs; p.do_hs = do_hs; p.dq_rs = dq_rs; p.dq_hs = dq_hs;
    p.dk_rs = dk_rs; p.dk_hs = dk_hs; p.dv_rs = dv_rs; p.dv_hs = dv_hs;
    p.lse_stride = lse_stride; p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    if (accum_f32 < 0 || accum_f32 > 2) return fail(LV_EINVAL, "accum_f32 = %d (0 bf16 results, 1 fp32 added to, 2 fp32 stored)", accum_f32);
    p.f32_store = (accum_f32 == 2);
    if (lv_launch_bwd(p, d, causal, accum_f32 != 0, (hipStream_t)stream))
        return fail(LV_ELAUNCH, "backward launch failed: %s", hipGetErrorString(hipGetLastError()));
    return LV_OK;
}

// ---------------------------------------------------------------------------
// ring merge: (out_acc, lse_acc) <- combine with (out_blk, lse_blk)
// one thread per 8 output elements
// ---------------------------------------------------------------------------
template <bool INIT>
__global__ void lv_merge_kernel(float* __restrict__ out_acc, float* __restrict__ lse_acc,
                                const __bf16* __restrict__ out_blk, const float* __restrict__ lse_blk,
                                int row0, int rows, int hq, int d, int64_t blk_rs, int64_t blk_hs, int64_t lse_stride) {
    const int cpr = d / 8;  // 8-element chunks per (row, head)
    int64_t total = (int64_t)rows * hq * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ch = (int)(idx % cpr);
        int64_t rh = idx / cpr;
        int head = (int)(rh % hq);
        int row = row0 + (int)(rh / hq);
        float lb = lse_blk[(int64_t)head * lse_stride + row];
        float* oa = out_acc + ((int64_t)row * hq + head) * d + ch * 8;
        const __bf16* ob = out_blk + (int64_t)row * blk_rs + (int64_t)head * blk_hs + ch * 8;
        bf16x8 b = *reinterpret_cast<const bf16x8*>(ob);
        f32x4 a0, a1;
        float ln;
        if constexpr (INIT) {   // the block IS the running result so far: the accumulators are never read
            ln = lb;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a0[i] = (float)b[i]; a1[i] = (float)b[i + 4]; }
        } else {
            float la = lse_acc[(int64_t)head * lse_stride + row];
            float m = fmaxf(la, lb);
            float wa = 0.f, wb = 0.f;
            ln = -INFINITY;
            if (m > -INFINITY) {
                float ea = __expf(la - m), eb = __expf(lb - m);
                float tot = ea + eb;
                wa = ea / tot; wb = eb / tot;
                ln = m + __logf(tot);
            }
            a0 = *reinterpret_cast<f32x4*>(oa); a1 = *reinterpret_cast<f32x4*>(oa + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a0[i] = a0[i] * wa + (float)b[i] * wb;
                a1[i] = a1[i] * wa + (float)b[i + 4] * wb;
            }
        }
        *reinterpret_cast<f32x4*>(oa) = a0;
        *reinterpret_cast<f32x4*>(oa + 4) = a1;
        // every chunk of a (row, head) computes the same lse; chunk 0 writes it after all
        // chunks of this (row, head) in this thread's wave have read la (same wave: cpr <= 16
        // consecutive threads), other waves never read this (row, head).
        if (ch == 0) lse_acc[(int64_t)head * lse_stride + row] = ln;
    }
}

static int merge_impl(bool init, float* out_acc, float* lse_acc, const void* out_blk, const float* lse_blk,
                      int row0, int rows, int hq, int d, int64_t blk_rs, int64_t blk_hs, int64_t lse_stride, void* stream) {
    if (!out_acc || !lse_acc || !out_blk || !lse_blk) return fail(LV_EINVAL, "null pointer");
    // the in-place lse update needs the d/8 threads of a (row, head) inside one wave: d/8 must divide 64
    if (d != 64 && d != 128) return fail(LV_EINVAL, "lvattn_merge: head dim %d not supported (64 or 128)", d);
    if (rows < 0 || hq <= 0 || row0 < 0 || lse_stride < (int64_t)row0 + rows) return fail(LV_EINVAL, "lvattn_merge: bad shape");
    if ((blk_rs | blk_hs) & 7) return fail(LV_EINVAL, "block strides must be multiples of 8");
    if (((uintptr_t)out_acc | (uintptr_t)out_blk) & 15) return fail(LV_EINVAL, "lvattn_merge: out_acc / out_blk must be 16-byte aligned");
    if (rows == 0) return LV_OK;
    int64_t total = (int64_t)rows * hq * (d / 8);
    int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
    if (init)
        hipLaunchKernelGGL(lv_merge_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out_acc, lse_acc,
                           (const __bf16*)out_blk, lse_blk, row0, rows, hq, d, blk_rs, blk_hs, lse_stride);
    else
        hipLaunchKernelGGL(lv_merge_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, out_acc, lse_acc,
                           (const __bf16*)out_blk, lse_blk, row0, rows, hq, d, blk_rs, blk_hs, lse_stride);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "merge launch failed");
}

extern "C" int lvattn_merge(float* out_acc, float* lse_acc, const void* out_blk, const float* lse_blk,
                            int row0, int rows, int hq, int d, int64_t blk_rs, int64_t blk_hs,
                            int64_t lse_stride, void* stream) {
    return merge_impl(false, out_acc, lse_acc, out_blk, lse_blk, row0, rows, hq, d, blk_rs, blk_hs, lse_stride, stream);
}

extern "C" int lvattn_merge_init(float* out_acc, float* lse_acc, const void* out_blk, const float* lse_blk,
                                 int row0, int rows, int hq, int d, int64_t blk_rs, int64_t blk_hs,
                                 int64_t lse_stride, void* stream) {
    return merge_impl(true, out_acc, lse_acc, out_blk, lse_blk, row0, rows, hq, d, blk_rs, blk_hs, lse_stride, stream);
}

// ---------------------------------------------------------------------------
// fp32 -> bf16 cast with output strides (dq_accum -> dq)
// ---------------------------------------------------------------------------
__global__ void lv_cast_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int64_t rows, int heads, int d,
                               int64_t rs, int64_t hs) {
    const int cpr = d / 8;
    int64_t total = rows * heads * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ch = (int)(idx % cpr);
        int64_t rh = idx / cpr;
        int head = (int)(rh % heads);
        int64_t row = rh / heads;
        const float* s = src + idx * 8;
        f32x4 a0 = *reinterpret_cast<const f32x4*>(s), a1 = *reinterpret_cast<const f32x4*>(s + 4);
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = (__bf16)a0[i]; o[i + 4] = (__bf16)a1[i]; }
        *reinterpret_cast<bf16x8*>(dst + row * rs + (int64_t)head * hs + ch * 8) = o;
    }
}

extern "C" int lvattn_cast_f32_bf16(const float* src, void* dst, int rows, int heads, int d,
                                    int64_t rs, int64_t hs, void* stream) {
    if (!src || !dst) return fail(LV_EINVAL, "null pointer");
    if (d % 8 || ((rs | hs) & 7) || ((uintptr_t)dst & 15)) return fail(LV_EINVAL, "bad alignment");
    if (rows <= 0) return LV_OK;
    int64_t total = (int64_t)rows * heads * (d / 8);
    int grid = (int)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(lv_cast_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, (int64_t)rows, heads, d, rs, hs);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "cast launch failed");
}

// ---------------------------------------------------------------------------
// pixel shuffle (scale 0.5) fused with the class-token drop.
// Derived from the reference's view/permute chain: with the ViT grid indexed [w][h] (w = token / side, h = token % side)
//   out[n][w/2][h/2][ (w%2)*2c + (h%2)*c + ch ] = in[n][w][h][ch]
// i.e. output token (w/2)*(side/2) + (h/2), channel block (w%2)*2 + (h%2).
// Pure 16-byte moves: one thread per 8 bf16.
// ---------------------------------------------------------------------------
__global__ void lv_pixel_shuffle_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int n, int side, int c,
                                        int skip, int64_t in_image_stride) {
    const int cpr = c / 8;
    const int half = side / 2;
    int64_t total = (int64_t)n * side * side * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int ch = (int)(idx % cpr);
        int64_t t = idx / cpr;
        int hh = (int)(t % side); t /= side;
        int ww = (int)(t % side);
        int img = (int)(t / side);
        const __bf16* src = in + (int64_t)img * in_image_stride + ((int64_t)(skip + ww * side + hh)) * c + ch * 8;
        int64_t otok = (int64_t)(ww / 2) * half + (hh / 2);
        __bf16* dst = out + (((int64_t)img * half * half + otok) * 4 + ((ww & 1) * 2 + (hh & 1))) * c + ch * 8;
        *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(src);
    }
}

extern "C" int lv_pixel_shuffle(const void* in, void* out, int n, int side, int c, int skip,
                                int64_t in_image_stride, void* stream) {
    if (!in || !out) return fail(LV_EINVAL, "null pointer");
    if (side <= 0 || (side & 1) || c % 8 || n < 0 || skip < 0) return fail(LV_EINVAL, "bad shape (side must be even, c %% 8 == 0)");
    if ((in_image_stride & 7) || (((uintptr_t)in | (uintptr_t)out) & 15)) return fail(LV_EINVAL, "bad alignment");
    if (((int64_t)skip * c) & 7) return fail(LV_EINVAL, "skip*c must be a multiple of 8");
    if (n == 0) return LV_OK;
    int64_t total = (int64_t)n * side * side * (c / 8);
    int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(lv_pixel_shuffle_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, (__bf16*)out, n, side, c,
                       skip, in_image_stride);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "pixel_shuffle launch failed");
}

// ---------------------------------------------------------------------------
// pixel shuffle, stride form (pixel_shuffle_v2): every st x st patch of the token grid becomes ONE output token whose
// channels are the patch's tokens in (row, column) order,
//   out[n][a * ceil(side/st) + b][ (di*st + dj)*c + ch ] = in[n][a*st + di][b*st + dj][ch]
// with zeros where the patch hangs over the edge of the grid (the reference pads right and bottom).  One thread per
// 16 bytes of output.
// ---------------------------------------------------------------------------
__global__ void lv_pixel_shuffle_v2_kernel(const __bf16* __restrict__ in, __bf16* __restrict__ out, int n, int side, int c,
                                           int skip, int st, int64_t in_image_stride) {
    const int cpr = c / 8;
    const int os = (side + st - 1) / st;
    const int64_t total = (int64_t)n * os * os * st * st * cpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % cpr);
        int64_t t = idx / cpr;
        const int dj = (int)(t % st); t /= st;
        const int di = (int)(t % st); t /= st;
        const int b = (int)(t % os); t /= os;
        const int a = (int)(t % os);
        const int img = (int)(t / os);
        const int i = a * st + di, j = b * st + dj;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (i < side && j < side)
            val = *reinterpret_cast<const u32x4*>(in + (int64_t)img * in_image_stride + ((int64_t)(skip + i * side + j)) * c + ch * 8);
        *reinterpret_cast<u32x4*>(out + idx * 8) = val;
    }
}

extern "C" int lv_pixel_shuffle_v2(const void* in, void* out, int n, int side, int c, int skip, int scale_stride,
                                   int64_t in_image_stride, void* stream) {
    if (!in || !out) return fail(LV_EINVAL, "null pointer");
    if (side <= 0 || scale_stride <= 0 || c <= 0 || c % 8 || n < 0 || skip < 0) return fail(LV_EINVAL, "lv_pixel_shuffle_v2: bad shape (c %% 8 == 0, stride > 0)");
    if ((in_image_stride & 7) || (((uintptr_t)in | (uintptr_t)out) & 15)) return fail(LV_EINVAL, "bad alignment");
    if (((int64_t)skip * c) & 7) return fail(LV_EINVAL, "skip*c must be a multiple of 8");
    if (n == 0) return LV_OK;
    const int os = (side + scale_stride - 1) / scale_stride;
    const int64_t total = (int64_t)n * os * os * scale_stride * scale_stride * (c / 8);
    int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(lv_pixel_shuffle_v2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, (__bf16*)out, n, side, c,
                       skip, scale_stride, in_image_stride);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "pixel_shuffle_v2 launch failed");
}

// ---------------------------------------------------------------------------
// row scatter / gather (16-byte or 4-byte granules)
// ---------------------------------------------------------------------------
template <typename VEC>
__global__ void lv_move_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, const int64_t* __restrict__ src_index,
                                    const int64_t* __restrict__ dst_index, int64_t n, int64_t row_bytes, int64_t src_rows,
                                    int64_t dst_rows) {
    const int64_t vpr = row_bytes / sizeof(VEC);
    int64_t total = n * vpr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t i = idx / vpr, e = idx % vpr;
        int64_t s = src_index ? src_index[i] : i;
        int64_t t = dst_index ? dst_index[i] : i;
        if (s < 0) s += src_rows;  // torch-style negative indices
        if (t < 0) t += dst_rows;
        if (s < 0 || s >= src_rows || t < 0 || t >= dst_rows) continue;  // out-of-range rows are skipped
        reinterpret_cast<VEC*>(dst + t * row_bytes)[e] = reinterpret_cast<const VEC*>(src + s * row_bytes)[e];
    }
}

static int move_rows(const void* src, void* dst, const int64_t* si, const int64_t* di, int64_t n, int64_t row_bytes,
                     int64_t src_rows, int64_t dst_rows, void* stream) {
    if (!src || !dst) return fail(LV_EINVAL, "null pointer");
    if (n < 0 || row_bytes <= 0 || (row_bytes & 3)) return fail(LV_EINVAL, "row bytes must be a positive multiple of 4");
    if (n == 0) return LV_OK;
    bool wide = !(row_bytes & 15) && !(((uintptr_t)src | (uintptr_t)dst) & 15);
    int64_t total = n * (row_bytes / (wide ? 16 : 4));
    int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    if (wide)
        hipLaunchKernelGGL(lv_move_rows_kernel<u32x4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, si, di, n,
                           row_bytes, src_rows, dst_rows);
    else
        hipLaunchKernelGGL(lv_move_rows_kernel<uint32_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src, (char*)dst, si, di,
                           n, row_bytes, src_rows, dst_rows);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "row move launch failed");
}

extern "C" int lv_scatter_rows(const void* src, void* dst, const int64_t* src_index, const int64_t* dst_index,
                               int64_t n, int width, int64_t src_rows, int64_t dst_rows, void* stream) {
    if (!dst_index) return fail(LV_EINVAL, "dst_index is null");
    if (width <= 0 || (width & 1)) return fail(LV_EINVAL, "width must be a positive even number of bf16 elements");
    return move_rows(src, dst, src_index, dst_index, n, (int64_t)width * 2, src_rows, dst_rows, stream);
}

extern "C" int lv_gather_rows(const void* src, void* dst, const int64_t* index, int64_t n,
                              int64_t row_bytes, int64_t src_rows, void* stream) {
    if (!index) return fail(LV_EINVAL, "index is null");
    return move_rows(src, dst, index, nullptr, n, row_bytes, src_rows, n, stream);
}

// ---------------------------------------------------------------------------
// shader-clock probe: what clock does the chip hold under a dense bf16 MFMA loop on the caller's (random) data?
// Every wave stamps the shader-clock counter and the constant 100 MHz counter around its loop;
// clock = d(shader counter) / d(100 MHz counter) * 100 MHz.  Diagnostic only: no product kernel carries stamps.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lv_clock_probe_kernel(const __bf16* __restrict__ seed, uint64_t* __restrict__ stamps,
                                                            float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    bf16x8 a = *reinterpret_cast<const bf16x8*>(seed + 8 * lane);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(seed + 8 * (64 + lane));
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    float keep = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) keep += acc[j][0] + acc[j][15];
    asm volatile("" : "+v"(keep));
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        stamps[2 * wave] = t1 - t0;
        stamps[2 * wave + 1] = r1 - r0;
        sink[wave] = keep;
    }
}

extern "C" int lv_clock_probe(const void* seed, uint64_t* stamps, float* sink, int n_workgroups, int iters, void* stream) {
    if (!seed || !stamps || !sink) return fail(LV_EINVAL, "lv_clock_probe: null pointer");
    if (((uintptr_t)seed & 15) || n_workgroups <= 0 || iters <= 0) return fail(LV_EINVAL, "lv_clock_probe: bad argument");
    hipLaunchKernelGGL(lv_clock_probe_kernel, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream, (const __bf16*)seed, stamps, sink, iters);
    return hipGetLastError() == hipSuccess ? LV_OK : fail(LV_ELAUNCH, "lv_clock_probe: launch failed");
}

// ---------------------------------------------------------------------------
// rotary position embedding
// ---------------------------------------------------------------------------
extern "C" int lv_rope(const void* x, void* out, const void* cos_tab, const void* sin_tab, const int32_t* pos_ids,
                       int64_t rows, int heads, int d, int rot, int64_t x_row_stride, int64_t x_head_stride,
                       int64_t o_row_stride, int64_t o_head_stride, int64_t tab_rows, int tab_div, int conj, void* stream) {
    if (!x || !out || !cos_tab || !sin_tab) return fail(LV_EINVAL, "lv_rope: null pointer");
    if (rows < 0 || heads <= 0) return fail(LV_EINVAL, "lv_rope: rows=%lld heads=%d", (long long)rows, heads);
    if (d <= 0 || d % 8 || rot <= 0 || rot > d || rot % 16)
        return fail(LV_EINVAL, "lv_rope: d=%d rot=%d (need d %% 8 == 0, rot %% 16 == 0, 0 < rot <= d)", d, rot);
    if (tab_rows <= 0 || tab_div <= 0) return fail(LV_EINVAL, "lv_rope: tab_rows=%lld tab_div=%d", (long long)tab_rows, tab_div);
    if (!pos_ids && rows > 0 && (rows - 1) / tab_div >= tab_rows)
        return fail(LV_EINVAL, "lv_rope: %lld rows / %d need more than the table's %lld rows", (long long)rows, tab_div, (long long)tab_rows);
    if (x_row_stride % 8 || x_head_stride % 8 || o_row_stride % 8 || o_head_stride % 8 ||
        ((uintptr_t)x | (uintptr_t)out | (uintptr_t)cos_tab | (uintptr_t)sin_tab) % 16)
        return fail(LV_EINVAL, "lv_rope: rows, heads and tables must be 16-byte aligned");
    int rc = lv_launch_rope(x, out, cos_tab, sin_tab, pos_ids, rows, heads, d, rot, x_row_stride, x_head_stride, o_row_stride,
                            o_head_stride, tab_rows, tab_div, conj, (hipStream_t)stream);
    return rc ? fail(LV_ELAUNCH, "lv_rope: launch failed") : LV_OK;
}

extern "C" int lv_rope_pack_kv(const void* k, const void* v, void* kv, const void* cos_tab, const void* sin_tab,
                               const int32_t* pos_ids, int64_t rows, int heads, int d, int rot,
                               int64_t k_row_stride, int64_t k_head_stride, int64_t v_row_stride, int64_t v_head_stride,
                               int64_t tab_rows, int tab_div, void* stream) {
    if (!k || !v || !kv || !cos_tab || !sin_tab) return fail(LV_EINVAL, "lv_rope_pack_kv: null pointer");
    if (rows < 0 || heads <= 0) return fail(LV_EINVAL, "lv_rope_pack_kv: rows=%lld heads=%d", (long long)rows, heads);
    if (d <= 0 || d % 8 || rot <= 0 || rot > d || rot % 16)
        return fail(LV_EINVAL, "lv_rope_pack_kv: d=%d rot=%d (need d %% 8 == 0, rot %% 16 == 0, 0 < rot <= d)", d, rot);
    if (tab_rows <= 0 || tab_div <= 0) return fail(LV_EINVAL, "lv_rope_pack_kv: tab_rows=%lld tab_div=%d", (long long)tab_rows, tab_div);
    if (!pos_ids && rows > 0 && (rows - 1) / tab_div >= tab_rows)
        return fail(LV_EINVAL, "lv_rope_pack_kv: %lld rows / %d need more than the table's %lld rows", (long long)rows, tab_div, (long long)tab_rows);
    if (k_row_stride % 8 || k_head_stride % 8 || v_row_stride % 8 || v_head_stride % 8 ||
        ((uintptr_t)k | (uintptr_t)v | (uintptr_t)kv | (uintptr_t)cos_tab | (uintptr_t)sin_tab) % 16)
        return fail(LV_EINVAL, "lv_rope_pack_kv: rows, heads and tables must be 16-byte aligned");
    int rc = lv_launch_rope_pack_kv(k, v, kv, cos_tab, sin_tab, pos_ids, rows, heads, d, rot, k_row_stride, k_head_stride,
                                    v_row_stride, v_head_stride, tab_rows, tab_div, (hipStream_t)stream);
    return rc ? fail(LV_ELAUNCH, "lv_rope_pack_kv: launch failed") : LV_OK;
}

// ---------------------------------------------------------------------------
// InternViT encoder pieces around the GEMMs
// ---------------------------------------------------------------------------
static bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

extern "C" int lv_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int h,
                            int64_t x_row_stride, int64_t y_row_stride, float eps, void* stream) {
    if (!x || !y || !gamma || !beta) return fail(LV_EINVAL, "lv_layernorm: null pointer");
    if (rows < 0 || h <= 0 || h % 8 || h > 8192) return fail(LV_EINVAL, "lv_layernorm: rows=%lld h=%d (need h %% 8 == 0, h <= 8192)", (long long)rows, h);
    if (x_row_stride % 8 || y_row_stride % 8 || !al16(x) || !al16(y) || !al16(gamma) || !al16(beta))
        return fail(LV_EINVAL, "lv_layernorm: rows and parameters must be 16-byte aligned");
    return lv_launch_layernorm(x, y, gamma, beta, rows, h, x_row_stride, y_row_stride, eps, (hipStream_t)stream)
               ? fail(LV_ELAUNCH, "lv_layernorm: launch failed") : LV_OK;
}

extern "C" int lv_bias_gelu(const void* x, const void* bias, void* y, int64_t rows, int h, void* stream) {
    if (!x || !y) return fail(LV_EINVAL, "lv_bias_gelu: null pointer");
    if (rows < 0 || h <= 0 || h % 8 || !al16(x) || !al16(y) || (bias && !al16(bias))) return fail(LV_EINVAL, "lv_bias_gelu: h=%d must be a multiple of 8, pointers 16-byte aligned", h);
    return lv_launch_bias_gelu(x, bias, y, rows, h, (hipStream_t)stream) ? fail(LV_ELAUNCH, "lv_bias_gelu: launch failed") : LV_OK;
}

extern "C" int lv_bias_gelu_bwd(const void* x, const void* bias, const void* dy, void* dx, int64_t rows, int h, void* stream) {
    if (!x || !dy || !dx) return fail(LV_EINVAL, "lv_bias_gelu_bwd: null pointer");
    if (rows < 0 || h <= 0 || h % 8 || !al16(x) || !al16(dy) || !al16(dx) || (bias && !al16(bias)))
        return fail(LV_EINVAL, "lv_bias_gelu_bwd: h=%d must be a multiple of 8, pointers 16-byte aligned", h);
    return lv_launch_bias_gelu_bwd(x, bias, dy, dx, rows, h, (hipStream_t)stream) ? fail(LV_ELAUNCH, "lv_bias_gelu_bwd: launch failed") : LV_OK;
}

extern "C" int64_t lv_layernorm_bwd_workspace_bytes(int64_t rows, int h) {
    if (rows < 0 || h <= 0) return 0;
    return (int64_t)lv_layernorm_bwd_parts(rows) * 2 * h * (int64_t)sizeof(float);
}

extern "C" int lv_layernorm_bwd(const void* x, const void* dy, const void* gamma, void* dx, float* dgamma_dbeta, void* workspace,
                                int64_t workspace_bytes, int64_t rows, int h, int64_t x_row_stride, int64_t dy_row_stride,
                                int64_t dx_row_stride, float eps, void* stream) {
    if (!x || !dy || !gamma || !dx || !dgamma_dbeta || !workspace) return fail(LV_EINVAL, "lv_layernorm_bwd: null pointer");
    if (rows <= 0 || h <= 0 || h % 8 || h > 4096)
        return fail(LV_EINVAL, "lv_layernorm_bwd: rows=%lld h=%d (need rows > 0, h %% 8 == 0, h <= 4096)", (long long)rows, h);
    if (x_row_stride % 8 || dy_row_stride % 8 || dx_row_stride % 8 || !al16(x) || !al16(dy) || !al16(dx) || !al16(gamma) ||
        !al16(dgamma_dbeta) || !al16(workspace))
        return fail(LV_EINVAL, "lv_layernorm_bwd: rows, parameters and workspace must be 16-byte aligned");
    if (workspace_bytes < lv_layernorm_bwd_workspace_bytes(rows, h))
        return fail(LV_EINVAL, "lv_layernorm_bwd: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                    (long long)lv_layernorm_bwd_workspace_bytes(rows, h));
    return lv_launch_layernorm_bwd(x, dy, gamma, dx, dgamma_dbeta, (float*)workspace, rows, h, x_row_stride, dy_row_stride,
                                   dx_row_stride, eps, (hipStream_t)stream)
               ? fail(LV_ELAUNCH, "lv_layernorm_bwd: launch failed") : LV_OK;
}

extern "C" int lv_bias_scale_residual(const void* x, const void* bias, const void* ls, const void* residual, void* out, int64_t rows,
                                      int h, void* stream) {
    if (!x || !ls || !residual || !out) return fail(LV_EINVAL, "lv_bias_scale_residual: null pointer");
    if (rows < 0 || h <= 0 || h % 8 || !al16(x) || !al16(out) || !al16(ls) || !al16(residual) || (bias && !al16(bias)))
        return fail(LV_EINVAL, "lv_bias_scale_residual: h=%d must be a multiple of 8, pointers 16-byte aligned", h);
    return lv_launch_bias_scale_residual(x, bias, ls, residual, out, rows, h, (hipStream_t)stream)
               ? fail(LV_ELAUNCH, "lv_bias_scale_residual: launch failed") : LV_OK;
}

extern "C" int lv_bias_scale_residual_layernorm(const void* x, const void* bias, const void* ls, const void* residual, void* out,
                                                void* normed, const void* gamma, const void* beta, int64_t rows, int h, float eps,
                                                void* stream) {
    if (!x || !ls || !residual || !out || !normed || !gamma || !beta) return fail(LV_EINVAL, "lv_bias_scale_residual_layernorm: null pointer");
    if (rows < 0 || h <= 0 || h % 8 || h > 8192)
        return fail(LV_EINVAL, "lv_bias_scale_residual_layernorm: rows=%lld h=%d (need h %% 8 == 0, h <= 8192)", (long long)rows, h);
    if (!al16(x) || !al16(out) || !al16(normed) || !al16(ls) || !al16(residual) || !al16(gamma) || !al16(beta) || (bias && !al16(bias)))
        return fail(LV_EINVAL, "lv_bias_scale_residual_layernorm: pointers must be 16-byte aligned");
    return lv_launch_bsr_layernorm(x, bias, ls, residual, out, normed, gamma, beta, rows, h, eps, (hipStream_t)stream)
               ? fail(LV_ELAUNCH, "lv_bias_scale_residual_layernorm: launch failed") : LV_OK;
}

extern "C" int lv_patchify(const void* img, void* patches, int n, int c, int height, int width, int patch, int kpad, void* stream) {
    if (!img || !patches) return fail(LV_EINVAL, "lv_patchify: null pointer");
    if (n < 0 || c <= 0 || patch <= 0 || height % patch || width % patch || kpad < c * patch * patch)
        return fail(LV_EINVAL, "lv_patchify: n=%d c=%d %dx%d patch=%d kpad=%d", n, c, height, width, patch, kpad);
    return lv_launch_patchify(img, patches, n, c, height, width, patch, kpad, (hipStream_t)stream) ? fail(LV_ELAUNCH, "lv_patchify: launch failed") : LV_OK;
}

extern "C" int lv_embed_tokens(const void* proj, const void* conv_bias, const void* class_token, const void* pos_emb, void* tokens,
                               int n, int n_patches, int n_class, int h, void* stream) {
    if (!proj || !conv_bias || !pos_emb || !tokens || (n_class > 0 && !class_token)) return fail(LV_EINVAL, "lv_embed_tokens: null pointer");
    if (n < 0 || n_patches <= 0 || n_class < 0 || h <= 0 || h % 8) return fail(LV_EINVAL, "lv_embed_tokens: n=%d patches=%d class=%d h=%d", n, n_patches, n_class, h);
    if (!al16(proj) || !al16(conv_bias) || !al16(pos_emb) || !al16(tokens) || (class_token && !al16(class_token)))
        return fail(LV_EINVAL, "lv_embed_tokens: pointers must be 16-byte aligned");
    return lv_launch_embed_tokens(proj, conv_bias, class_token, pos_emb, tokens, n, n_patches, n_class, h, (hipStream_t)stream)
               ? fail(LV_ELAUNCH, "lv_embed_tokens: launch failed") : LV_OK;
}
