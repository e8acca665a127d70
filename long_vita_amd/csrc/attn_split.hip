// Split sweeps: filling the chip when a launch has fewer work items than compute units.
//
// A forward (or dQ) work item is one 256-row query block of ONE head sweeping all its keys; a dK/dV work item one
// 128-key block of one KV head sweeping all its query rows.  At the reference's per-device head count (tensor parallel
// 8: five query heads and one KV head per device, scripts/modellink/qwen25/finetune_qwen25_14b_intern_300m_ptd_tp8pp1_
// stage1.sh:101) a packed batch of a few thousand tokens is a few dozen items on 256 compute units, the longest of them
// (the last block of a causal segment) many times the average.  Here such sweeps are cut into pieces: every (block,
// piece) pair becomes a work item of its own that writes its partial result to workspace rows of its own (descriptor
// field out_delta), and one small kernel combines a block's pieces IN A FIXED ORDER -- log-sum-exp merge for the
// forward's (out, lse), plain sums for the fp32 gradient partials -- so results stay bitwise reproducible from run to
// run, as everywhere in this library (no atomics).  Host side: a planner that picks the piece length from a simulated
// schedule, and the table builders; device side: the two combine kernels.
#include "internal.h"
#include "../../include/lvattn.h"
#include <algorithm>
#include <cstring>
#include <queue>
#include <vector>

namespace {

constexpr int FWD_BLOCK = 256;
constexpr int BWD_BLOCK = LV_BWD_BLOCK_N_INTERNAL;

struct Item { int a0, an, b0, bn, shift, block; };   // a: the block's rows, b: the swept range (piece), block: index of the base block

// base blocks of a segment list, forward / dQ form: (q rows, visible key range, shift)
void fwd_base(const int32_t* cu, int n_seg, int causal, std::vector<Item>& out) {
    for (int s = 0; s < n_seg; ++s) {
        const int q0 = cu[s], ql = cu[s + 1] - cu[s];
        for (int m = 0; m < ql; m += FWD_BLOCK) {
            const int rows = std::min(FWD_BLOCK, ql - m);
            const int sh = causal ? m : 0x3fffffff;
            const int klen = causal ? std::min(ql, rows + m) : ql;
            out.push_back({q0 + m, rows, q0, klen, sh, (int)out.size()});
        }
    }
}
// dK/dV form: (key rows, query range that sees them, shift); the query range starts on a multiple of 32 rows
void bwd_base(const int32_t* cu, int n_seg, int causal, std::vector<Item>& out) {
    for (int s = 0; s < n_seg; ++s) {
        const int k0 = cu[s], kl = cu[s + 1] - cu[s];
        for (int n = 0; n < kl; n += BWD_BLOCK) {
            const int rows = std::min(BWD_BLOCK, kl - n);
            const int qstart = causal ? (std::min(n, kl) & ~31) : 0;
            const int sh = causal ? qstart - n : 0x3fffffff;
            out.push_back({k0 + n, rows, k0 + qstart, kl - qstart, sh, (int)out.size()});
        }
    }
}

// pieces of one base block: the swept range cut every `piece` elements; a piece's `shift` field carries its cut position
// (offset of its first swept element inside the block's sweep), which the table builder turns into the piece's own shift
void cut(const Item& b, int piece, std::vector<Item>& out) {
    if (b.bn <= 0) {
        Item it = b;
        it.shift = 0;
        out.push_back(it);
        return;
    }
    for (int c = 0; c < b.bn; c += piece) {
        Item it = b;
        it.b0 = b.b0 + c;
        it.bn = std::min(piece, b.bn - c);
        it.shift = c;
        out.push_back(it);
    }
}

// makespan of a list of costs on n_slots machines, heaviest first, each to the least loaded slot (what the dispatcher of
// a launch ordered heaviest-first amounts to)
double makespan(std::vector<double>& cost, int n_slots) {
    std::sort(cost.begin(), cost.end(), std::greater<double>());
    std::priority_queue<double, std::vector<double>, std::greater<double>> load;
    for (int i = 0; i < n_slots; ++i) load.push(0.0);
    double worst = 0.0;
    for (double c : cost) {
        double l = load.top() + c;
        load.pop();
        load.push(l);
        worst = std::max(worst, l);
    }
    return worst;
}

// piece length (0 = leave the sweeps whole) for base blocks with sweep lengths bn, `copies` identical copies of every
// item (heads), per-item fixed cost `fixed` in sweep elements
int plan(const std::vector<Item>& base, int copies, int n_cu, int fixed, int min_piece) {
    if (base.empty() || (int64_t)base.size() * copies >= 2LL * n_cu) return 0;
    auto span = [&](int piece) {
        std::vector<double> cost;
        for (const Item& b : base)
            for (int c = 0; c < std::max(1, b.bn); c += piece)
                for (int h = 0; h < copies; ++h) cost.push_back((double)std::min(piece, std::max(1, b.bn) - c) + fixed);
        return makespan(cost, n_cu);
    };
    const double whole = span(1 << 30);
    int best_piece = 0;
    double best = whole;
    for (int piece = 32768; piece >= min_piece; piece >>= 1) {
        const double m = span(piece);
        if (m < best * 0.97) { best = m; best_piece = piece; }   // shorter pieces only for a gain of more than 3 %
    }
    return (best <= 0.9 * whole) ? best_piece : 0;               // under a tenth to gain: not worth the workspace and the combine pass
}

int check_cu_list(const int32_t* cu, int n_seg) {
    if (!cu || n_seg < 0) return -1;
    for (int i = 0; i < n_seg; ++i)
        if (cu[i + 1] < cu[i] || cu[0] < 0) return -1;
    return 0;
}

// the split tables: descriptors (heaviest first) + combine entries (out_row0, rows, ws_row0, n_pieces)
int build(std::vector<Item>& base, int piece, bool causal, bool backward, int32_t* desc, int max_blocks, int32_t* comb, int max_comb,
          int* n_comb, int* ws_rows) {
    const int BLK = backward ? BWD_BLOCK : FWD_BLOCK;
    struct Out { int32_t v[8]; int64_t work; };
    std::vector<Out> items;
    int ws = 0, nc = 0;
    for (const Item& b : base) {
        std::vector<Item> pieces;
        cut(b, piece, pieces);
        if (comb) {
            if (nc >= max_comb) return -3;
            comb[4 * nc + 0] = b.a0; comb[4 * nc + 1] = b.an; comb[4 * nc + 2] = ws; comb[4 * nc + 3] = (int)pieces.size();
        }
        ++nc;
        for (const Item& pc : pieces) {
            Out o{};
            const int c = pc.shift;                    // cut position inside the base sweep
            // forward / dQ: key j' = j - c of the piece is visible to row i iff j' <= i + shift - c
            // dK/dV:        key jj is visible to row ii' = ii - c of the piece iff jj <= ii' + shift + c
            int sh = b.shift;
            if (causal) sh = backward ? b.shift + c : b.shift - c;
            o.v[0] = pc.a0; o.v[1] = pc.an; o.v[2] = pc.b0; o.v[3] = pc.bn; o.v[4] = sh; o.v[5] = ws - pc.a0;
            if (!backward && causal) {                 // keys of the piece no row of the block sees are not swept
                const int64_t vis = (int64_t)pc.an + sh;
                o.v[3] = (int)std::max<int64_t>(0, std::min<int64_t>(pc.bn, vis));
            }
            o.work = (int64_t)o.v[1] * o.v[3];
            items.push_back(o);
            ws += BLK;
        }
    }
    if (n_comb) *n_comb = nc;
    if (ws_rows) *ws_rows = ws;
    if (desc) {
        if ((int)items.size() > max_blocks) return -3;
        std::stable_sort(items.begin(), items.end(), [](const Out& a, const Out& b) { return a.work > b.work; });
        for (size_t i = 0; i < items.size(); ++i) memcpy(desc + 8 * i, items[i].v, sizeof(int32_t) * 8);
    }
    return (int)items.size();
}

}  // namespace

extern "C" int lv_plan_fwd_split(const int32_t* cu, int n_seg, int causal, int heads, int n_cu) {
    if (check_cu_list(cu, n_seg) || heads <= 0 || n_cu <= 0) return LV_EINVAL;
    std::vector<Item> base;
    fwd_base(cu, n_seg, causal, base);
    // fixed cost of a forward / dQ work item, in keys of sweep: prologue and epilogue of the workgroup plus its share of the
    // workspace round trip and the combine pass.  Measured at 5 / 1 heads: 8,192 tokens run best with 4,096 keys per piece
    // (563 TFLOP/s forward + backward against 527 with 1,024), 16,384 tokens with 8,192 (762 against 722 with 2,048).
    return plan(base, heads, n_cu, 512, 256);
}
extern "C" int lv_plan_bwd_split(const int32_t* cu, int n_seg, int causal, int kv_heads, int group, int n_cu) {
    if (check_cu_list(cu, n_seg) || kv_heads <= 0 || group <= 0 || n_cu <= 0) return LV_EINVAL;
    std::vector<Item> base;
    bwd_base(cu, n_seg, causal, base);
    // a key block's sweep is `group` heads long per query row; its fixed cost (K / V fragments, the ring's prologue, the
    // epilogue) is worth about 128 rows of one head
    return plan(base, kv_heads, n_cu, (128 + group - 1) / group, 256);
}
extern "C" int lv_build_fwd_blocks_split(const int32_t* cu, int n_seg, int causal, int piece_keys, int32_t* desc, int max_blocks,
                                         int32_t* comb, int max_comb, int* n_comb, int* ws_rows) {
    if (check_cu_list(cu, n_seg) || piece_keys <= 0 || piece_keys % 64) return LV_EINVAL;
    std::vector<Item> base;
    fwd_base(cu, n_seg, causal, base);
    return build(base, piece_keys, causal != 0, false, desc, max_blocks, comb, max_comb, n_comb, ws_rows);
}
extern "C" int lv_build_bwd_blocks_split(const int32_t* cu, int n_seg, int causal, int piece_rows, int32_t* desc, int max_blocks,
                                         int32_t* comb, int max_comb, int* n_comb, int* ws_rows) {
    if (check_cu_list(cu, n_seg) || piece_rows <= 0 || piece_rows % 32) return LV_EINVAL;
    std::vector<Item> base;
    bwd_base(cu, n_seg, causal, base);
    return build(base, piece_rows, causal != 0, true, desc, max_blocks, comb, max_comb, n_comb, ws_rows);
}

// ---------------------------------------------------------------------------
// combine kernels: grid (entry x row chunk, head); a workgroup handles COMB_ROWS rows of one block and head, one thread
// per 8 output elements (launches of this path are small: the row chunks are what fills the chip)
// ---------------------------------------------------------------------------
constexpr int COMB_ROWS = 32;

__global__ __launch_bounds__(256) void lv_combine_lse_kernel(const __bf16* __restrict__ ws_out, const float* __restrict__ ws_lse,
                                                             __bf16* __restrict__ out, float* __restrict__ lse,
                                                             const int32_t* __restrict__ comb, int blk, int hq, int d,
                                                             int64_t ws_lse_stride, int64_t o_rs, int64_t o_hs, int64_t lse_stride) {
    const int chunks = blk / COMB_ROWS;
    const int e = blockIdx.x / chunks, head = blockIdx.y;
    const int out_row0 = comb[4 * e], rows = comb[4 * e + 1], ws_row0 = comb[4 * e + 2], np = comb[4 * e + 3];
    const int cpr = d / 8;
    const int r0 = (blockIdx.x % chunks) * COMB_ROWS, r1 = min(rows, r0 + COMB_ROWS);
    for (int idx = threadIdx.x; idx < (r1 - r0) * cpr; idx += blockDim.x) {
        const int row = r0 + idx / cpr, ch = idx % cpr;
        const float* lp = ws_lse + (int64_t)head * ws_lse_stride + ws_row0 + row;
        float m = -INFINITY;
        for (int p = 0; p < np; ++p) m = fmaxf(m, lp[(int64_t)p * blk]);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float tot = 0.f;
        if (m > -INFINITY) {
            for (int p = 0; p < np; ++p) {       // fixed order: piece 0 first
                const int64_t wr = ws_row0 + (int64_t)p * blk + row;
                const float w = __expf(lp[(int64_t)p * blk] - m);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(ws_out + (wr * hq + head) * d + ch * 8);
                tot += w;
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += w * (float)b[i];
            }
        }
        const float inv = tot > 0.f ? 1.f / tot : 0.f;
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)(acc[i] * inv);
        *reinterpret_cast<bf16x8*>(out + (int64_t)(out_row0 + row) * o_rs + (int64_t)head * o_hs + ch * 8) = o;
        if (ch == 0) lse[(int64_t)head * lse_stride + out_row0 + row] = tot > 0.f ? m + __logf(tot) : -INFINITY;
    }
}

__global__ __launch_bounds__(256) void lv_combine_sum_kernel(const float* __restrict__ ws, __bf16* __restrict__ out,
                                                             const int32_t* __restrict__ comb, int blk, int heads, int d, int64_t o_rs,
                                                             int64_t o_hs) {
    const int chunks = blk / COMB_ROWS;
    const int e = blockIdx.x / chunks, head = blockIdx.y;
    const int out_row0 = comb[4 * e], rows = comb[4 * e + 1], ws_row0 = comb[4 * e + 2], np = comb[4 * e + 3];
    const int cpr = d / 8;
    const int r0 = (blockIdx.x % chunks) * COMB_ROWS, r1 = min(rows, r0 + COMB_ROWS);
    for (int idx = threadIdx.x; idx < (r1 - r0) * cpr; idx += blockDim.x) {
        const int row = r0 + idx / cpr, ch = idx % cpr;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < np; ++p) {           // fixed order: piece 0 first
            const float* src = ws + ((ws_row0 + (int64_t)p * blk + row) * heads + head) * d + ch * 8;
            const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[i] += a[i]; acc[4 + i] += b[i]; }
        }
        bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)acc[i];
        *reinterpret_cast<bf16x8*>(out + (int64_t)(out_row0 + row) * o_rs + (int64_t)head * o_hs + ch * 8) = o;
    }
}

extern "C" int lvattn_combine_lse(const void* ws_out, const float* ws_lse, void* out, float* lse, const int32_t* comb_dev, int n_comb,
                                  int hq, int d, int64_t ws_lse_stride, int64_t o_row_stride, int64_t o_head_stride, int64_t lse_stride,
                                  void* stream) {
    if (!ws_out || !ws_lse || !out || !lse || !comb_dev) return LV_EINVAL;
    if (n_comb < 0 || hq <= 0 || hq > 65535 || (d != 64 && d != 128) || (o_row_stride | o_head_stride) & 7) return LV_EINVAL;
    if (n_comb == 0) return LV_OK;
    hipLaunchKernelGGL(lv_combine_lse_kernel, dim3(n_comb * (FWD_BLOCK / COMB_ROWS), hq), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)ws_out, ws_lse, (__bf16*)out, lse, comb_dev, FWD_BLOCK, hq, d, ws_lse_stride, o_row_stride,
                       o_head_stride, lse_stride);
    return hipGetLastError() == hipSuccess ? LV_OK : LV_ELAUNCH;
}
extern "C" int lvattn_combine_sum(const float* ws, void* out, const int32_t* comb_dev, int n_comb, int block_rows, int heads, int d,
                                  int64_t o_row_stride, int64_t o_head_stride, void* stream) {
    if (!ws || !out || !comb_dev) return LV_EINVAL;
    if (n_comb < 0 || heads <= 0 || heads > 65535 || (d != 64 && d != 128) || (o_row_stride | o_head_stride) & 7) return LV_EINVAL;
    if (block_rows != FWD_BLOCK && block_rows != BWD_BLOCK) return LV_EINVAL;
    if (n_comb == 0) return LV_OK;
    hipLaunchKernelGGL(lv_combine_sum_kernel, dim3(n_comb * (block_rows / COMB_ROWS), heads), dim3(256), 0, (hipStream_t)stream, ws,
                       (__bf16*)out, comb_dev, block_rows, heads, d, o_row_stride, o_head_stride);
    return hipGetLastError() == hipSuccess ? LV_OK : LV_ELAUNCH;
}
