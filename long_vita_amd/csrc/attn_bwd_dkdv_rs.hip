// dK / dV kernel of the attention backward for gfx950, "role-split" form.
//
// One workgroup = 8 waves = one 128-key block of one K/V head.  The two waves of a SIMD own the SAME
// 32 keys (key on the lane) and split the work on them:
//   V-wave (waves 0-3): S = Q K^T, P = exp2(c S - lse2) (masked, bf16), hands P to its partner through
//                       LDS, dV^T += dO^T P                    -> K fragments + dV^T accumulators in registers
//   K-wave (waves 4-7): dP = dO V^T, dS = P (dP - delta), dK^T += Q^T dS
//                                                              -> V fragments + dK^T accumulators in registers
// 16 MFMAs per 32-row query slice each.  The K-wave runs one slice behind the V-wave (it needs P), so
// one workgroup barrier per slice is enough, and on every SIMD one wave's vector work (exp / dS) sits
// beside the other's matrix work.  With the accumulators split, a wave needs < 256 registers, which is
// what lets two of them share a SIMD; no gradient is summed across workgroups.
//
// Q / dO / lse / delta of a slice arrive by direct global -> LDS loads into a 5-slot ring (slices
// it-1 .. it+3 around the V-wave's slice `it`), each wave fetching 1/8 of a slice plus its own copy of
// the row constants; a counted s_waitcnt vmcnt before the barrier leaves the newest request in flight.
// The sweep walks the query slices from the last one down (see attn_bwd.hip).
#include "internal.h"
#include <type_traits>
#include <cstdlib>

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_RS_MAP
#define LV_RS_MAP 1   // key block -> workgroup order: 0 = KV-head-major, 1 = KV head fastest (one head group per XCD)
#endif
#ifndef LV_RS_CFOLD
#define LV_RS_CFOLD 0   // 1: row constants ride in as the C operand of the S / dP products (see the kernel); measured equal, off
#endif
#ifndef LV_RS_DEPTH1
#define LV_RS_DEPTH1 2   // one unit per iteration: how many iterations ahead the staging requests run
#endif
#ifndef LV_RS_DMA_FIRST
#define LV_RS_DMA_FIRST 0   // staging requests at the start (1) or at the end (0) of the vector phase
#endif
#ifndef LV_RS_PRIO_M
#define LV_RS_PRIO_M 2   // s_setprio level in the matrix phase
#endif
#ifndef LV_RS_PRIO_V
#define LV_RS_PRIO_V 0   // s_setprio level in the vector phase
#endif

#ifdef LV_RS_STAMP
// Diagnostic build only: per role (V-wave / K-wave) sums of shader-clock ticks spent in the vector work, waiting at the
// barrier that ends it, in the matrix work, waiting at the barrier that ends that, plus iteration and wave counts.
__device__ unsigned long long lv_rs_stamps[2][8];
#define LV_STAMP(var) const uint64_t var = __builtin_amdgcn_s_memtime()
#else
#define LV_STAMP(var)
#endif

template <int D>
__device__ __forceinline__ int lds_off_rs(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

// Direct global -> LDS loads (lane i -> LDS byte lds_dst + 16 i or + 4 i; lds_dst in M0).  As asm so
// that the compiler does not count them; completion is waited for by hand.
__device__ __forceinline__ void rs_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void rs_dma4(const float* g, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

template <int D, bool CAUSAL, bool ACCUM_F32, int NSL>
__global__ __launch_bounds__(512) void lv_bwd_dkdv_rs_kernel(BwdParams p) {
    constexpr int NW = 8;
    constexpr int BQ = 32;             // query rows per unit (one 32-row slice of one head)
    constexpr int KS = D / 16;
    constexpr int MT = D / 32;
    constexpr int ROWB = D * 2;
    constexpr int CPR = D / 8;
    constexpr int QTILE = BQ * ROWB;   // bytes of a unit's Q (or dO) rows
    constexpr int DEPTH = (NSL == 1) ? LV_RS_DEPTH1 : 2;   // staging distance in iterations
    constexpr int NBUF = (DEPTH + 2) * NSL;   // ring slots: units of iterations it-1 .. it+DEPTH
    constexpr int AHEAD = DEPTH * NSL;        // staging distance in units
    constexpr int SLOTB = 2 * QTILE + 256;        // [Q rows][dO rows][32 lse, 32 delta f32]
    constexpr int RPP = 64 / CPR;                 // rows per 1-KiB piece
    constexpr int PPW = (2 * QTILE / 1024) / NW;  // pieces per wave per unit: 2 (d=128) or 1 (d=64)
    constexpr int PBUF = 4 * 2048;                // P of one unit: 4 wave pairs x (2 halves x 64 lanes x 16 bytes)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds0 = (lds_char*)smem;
    lds_char* const ldsP = lds0 + NBUF * SLOTB;   // [NSL][PBUF]
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_k = (w >= 4);        // K-wave (dK) or V-wave (dV)
    const int pair = w & 3;            // which 32 keys
    const int r = lane & 31;
    const int h = lane >> 5;

    const int bid = blockIdx.x;
#if LV_RS_MAP == 0
    const int blk = bid % p.n_blocks;
    const int kvh = bid / p.n_blocks;
#else
    // Workgroup i runs on XCD i % 8 (own L2 each).  Key blocks are dealt to the XCDs by KV head (KV head = unit % hkv:
    // with 8 KV heads one per XCD), so that an XCD's L2 only ever holds the Q / dO rows of one head group.
    const int unit = (bid >> 3) * 8 + (bid & 7);
    if (unit >= p.n_blocks * p.hkv) return;
    const int blk = unit / p.hkv;
    const int kvh = unit % p.hkv;
#endif

    const AttnDesc dsc = p.desc[blk];
    const int k_row0 = dsc.a_row0, k_rows = dsc.a_rows;
    const int q_row0 = dsc.b_row0, q_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;
    const int band_low = CAUSAL ? lv_band_low(dsc) : -(1 << 29);   // key kj visible to row ii only if kj >= ii + band_low
    const int nslice = (q_len + BQ - 1) / BQ;
    const int nunits = nslice * p.group;
    const int niter = (nunits + NSL - 1) / NSL;   // the last iteration may hold dead units (P forced to 0)

    // ---- this wave's B operand: K fragments (V-wave) or V fragments (K-wave) of key 32*pair + r
    const int kbase_w = 32 * pair;
    const int kj = kbase_w + r;
    bf16x8 bfr[KS];
    {
        const int grow = k_row0 + (kj < k_rows ? kj : k_rows - 1);
        const __bf16* bp = is_k ? p.v + (int64_t)grow * p.v_rs + (int64_t)kvh * p.v_hs + 8 * h
                                : p.k + (int64_t)grow * p.k_rs + (int64_t)kvh * p.k_hs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8*>(bp + 16 * ks);
    }
#if LV_RS_CFOLD
    // The V-wave needs P = 2^(c S - lse2), the K-wave dS = P (dP - delta).  Both additions leave the vector phase:
    // -lse2 (V-wave) / -delta (K-wave) of the unit's rows are the INITIAL accumulator of the S / dP product, and the
    // factor c = scale * log2(e) is folded into the V-wave's K fragments once (one more bf16 rounding of K in the
    // recomputed scores only: a relative 2^-9 on P, far inside the gradient tolerance; dK / dV use the unscaled data).
    if (!is_k) {
        const float cq = p.scale_log2e;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) bfr[ks][e] = (__bf16)((float)bfr[ks][e] * cq);
    }
#endif
    // retire these loads now: inside the loop the only memory counter traffic is the staging
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bfr[ks]));

    // ---- staging of a unit: waves 0-3 bring the Q rows, waves 4-7 the dO rows; piece j of wave w is
    // rows RPP*(PPW*(w&3) + j) .. +RPP-1.  The swizzle of the LDS image goes on the source.
    const int64_t x_rs = is_k ? p.do_rs : p.q_rs;
    const int64_t x_hs = is_k ? p.do_hs : p.q_hs;
    const __bf16* const x_base = (is_k ? p.dout : p.q) + (int64_t)q_row0 * x_rs;
    const int prow0 = RPP * PPW * pair + lane / CPR;
    const int rows_last = q_len - (nslice - 1) * BQ;  // only the last slice (the first one requested) can be partial
    int voff_full[PPW], voff_part[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        const int ch16 = lds_off_rs<D>(row, lane % CPR) - row * ROWB;
        voff_full[j] = row * (int)(x_rs * 2) + ch16;
        voff_part[j] = min(row, rows_last - 1) * (int)(x_rs * 2) + ch16;  // rows past the end repeat the last valid one
    }
    const int c_part = min(r, rows_last - 1) - r;
    int ld_sl = nslice - 1, ld_hg = 0, ld_n = 0;  // (slice, head) of the next request; requests so far
    uint32_t ld_slot = lds_base;
    const char* ld_x = reinterpret_cast<const char*>(x_base + (int64_t)kvh * p.group * x_hs + (int64_t)ld_sl * BQ * x_rs);
    const int64_t x_step_head = x_hs * 2;
    const int64_t x_step_slice = -((int64_t)(p.group - 1) * x_hs + (int64_t)BQ * x_rs) * 2;
    // row constants: lanes 0..31 fetch lse of rows 0..31, lanes 32..63 delta of rows 0..31, one request per
    // unit for the whole workgroup: V-wave `pair` brings them for the units u with u % NSL == pair (a
    // V-wave, because the V-waves run ahead: what they have waited for is there for both roles).  Each
    // direct-to-LDS request costs its wave (and its SIMD) far more issue time than an ordinary load, so
    // they are not duplicated.
    const bool c_wave = !is_k && pair < NSL;
    const float* ld_c = ((lane < 32) ? p.lse : p.delta) + q_row0 + (int64_t)kvh * p.group * p.lse_stride + (int64_t)ld_sl * BQ + r;
    const int64_t c_step_head = p.lse_stride;
    const int64_t c_step_slice = -((int64_t)(p.group - 1) * p.lse_stride + BQ);
    auto issue_dma = [&]() __attribute__((always_inline)) {
        const bool partial = (ld_sl == nslice - 1);
        const int rows_valid = partial ? rows_last : BQ;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ld_x), 0, (uint32_t)((rows_valid - 1) * x_rs * 2 + ROWB), 0x00020000);
        const uint32_t dst = ld_slot + (is_k ? QTILE : 0) + PPW * pair * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j) rs_dma16(rs, partial ? voff_part[j] : voff_full[j], dst + j * 1024);
        if (c_wave && (ld_n % NSL) == pair) rs_dma4(ld_c + (partial ? c_part : 0), ld_slot + 2 * QTILE);
        // past the last unit the stream stays where it is: the loop's unconditional requests re-fetch that
        // unit (into slots only dead units read)
        const bool adv = (++ld_n < nunits);
        const bool wrap = (ld_hg + 1 == p.group);
        const int64_t xs = wrap ? x_step_slice : x_step_head, cs = wrap ? c_step_slice : c_step_head;
        ld_x += adv ? xs : 0;
        ld_c += adv ? cs : 0;
        ld_hg = adv ? (wrap ? 0 : ld_hg + 1) : ld_hg;
        ld_sl = (adv && wrap) ? ld_sl - 1 : ld_sl;
        ld_slot = (ld_slot + SLOTB == lds_base + NBUF * SLOTB) ? lds_base : ld_slot + SLOTB;
    };

    // ---- LDS read addresses in slot 0.  The V-wave reads Q by rows and dO by columns, the K-wave dO by
    // rows and Q by columns: one pointer set each, pre-offset to the wave's two operands.
    const int row_op = is_k ? QTILE : 0;   // operand read by rows
    const int col_op = is_k ? 0 : QTILE;   // operand read by columns
    lds_char* a_ptr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_ptr[ks] = lds0 + row_op + lds_off_rs<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* t_ptr[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            t_ptr[mt][jh] = lds0 + col_op + lds_off_rs<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);
    // row constants: lse (V-wave) or delta (K-wave) of rows 8g+4h .. +3
    lds_char* const ld_ptr = lds0 + 2 * QTILE + 16 * h + (is_k ? 128 : 0);
    lds_char* const p_ptr = ldsP + pair * 2048 + lane * 16;   // this lane's P: halves at +0 and +1024

    f32x16 acc[MT];   // dV^T (V-wave) or dK^T (K-wave)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
    const float c = p.scale_log2e;

    // Both roles run the SAME loop, with role-dependent operand addresses, the K-waves one barrier
    // behind the V-waves.  An iteration handles NSL units u:
    //   VALU(it) : Y_u = f(X_u)    V-wave: P = exp2(c S - lse2), masked, also written to LDS for the partner
    //                              K-wave: dS = P (dP - delta), P read back from LDS
    //   MFMA(it) : X_u' = A_rows(unit of it+1) B^T   (V: S = Q K^T        K: dP = dO V^T)        8 MFMAs per unit
    //              acc += A_cols(unit of it)^T Y_u    (V: dV^T += dO^T P   K: dK^T += Q^T dS)     8 MFMAs per unit
    // so on every SIMD one wave is in its matrix phase while its partner is in its vector phase, and the
    // accumulators never cross a control-flow merge.  The LDS fragments of MFMA group g+1 are read while
    // group g runs.  More units per iteration = longer phases for the same two barriers.
    constexpr int NG = 4;               // MFMA groups per unit
    constexpr int NGT = NG * NSL;       // groups per matrix phase
    constexpr int XPG = KS / NG;        // X-chain steps per group: 2 or 1
    constexpr int APG = (2 * MT) / NG;  // accumulator products per group: 2 or 1
    static_assert(APG == XPG, "one accumulator product per X-chain step");
    bf16x8 rfr[2][XPG], cfr[2][APG];
    auto load_group = [&](int g, int set, int sb_x, int sb_a) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XPG; ++i) rfr[set][i] = *(LDS_PTR(bf16x8, a_ptr[XPG * g + i] + sb_x));
#pragma unroll
        for (int i = 0; i < APG; ++i) {
            const int j = APG * g + i, s2 = j / MT, mt = j % MT;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][0] + sb_a + 16 * s2 * ROWB));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][1] + sb_a + 16 * s2 * ROWB));
            bf16x8& af = cfr[set][i];
            af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
            af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
        }
    };

    if (niter > 0) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) issue_dma();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        f32x16 x[NSL];
        // row constants of a unit: -lse * log2(e) (V-wave) or -delta (K-wave) of rows 8g+4h .. +3
        const float cmul = is_k ? -1.f : -1.4426950408889634f;
#if LV_RS_CFOLD
        // ... read straight into the accumulator the unit's S / dP product starts from
        auto load_consts = [&](int u, int sbx) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t4 = *(LDS_PTR(f32x4, ld_ptr + sbx + 32 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) x[u][4 * g + e] = t4[e] * cmul;
            }
        };
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            load_consts(u, u * SLOTB);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fa = *(LDS_PTR(bf16x8, a_ptr[ks] + u * SLOTB));
                x[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, bfr[ks], x[u], 0, 0, 0);
            }
        }
#else
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[u][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 fa = *(LDS_PTR(bf16x8, a_ptr[ks] + u * SLOTB));
                x[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, bfr[ks], x[u], 0, 0, 0);
            }
        }
        // ... read one matrix phase ahead, so that the vector phase starts on registers, not on an LDS round trip
        f32x4 cst[NSL][4];
        auto load_consts = [&](int u, int sbx) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 t4 = *(LDS_PTR(f32x4, ld_ptr + sbx + 32 * g));
#pragma unroll
                for (int e = 0; e < 4; ++e) cst[u][g][e] = t4[e] * cmul;
            }
        };
#pragma unroll
        for (int u = 0; u < NSL; ++u) load_consts(u, u * SLOTB);
#endif
        bf16x8 y[NSL][2];
        int c_sl = nslice - 1, c_hg = 0;  // (slice, head) of the next unit
        int c_n = 0;                      // units consumed so far
        int sb = 0;                       // ring byte offset of this iteration's first unit
        if (is_k) asm volatile("s_barrier" ::: "memory");
#ifdef LV_RS_STAMP
        uint64_t st_vec = 0, st_bar1 = 0, st_mat = 0, st_bar2 = 0;
#endif
        for (int it = 0; it < niter; ++it) {
            const int sb_next = (sb + NSL * SLOTB == NBUF * SLOTB) ? 0 : sb + NSL * SLOTB;
            LV_STAMP(ta);
            // ================= VALU(it) =================
#if LV_RS_DMA_FIRST
            // units of iteration it+2 (their slots held iteration it-2, whose last reader finished two barriers ago),
            // requested while the last matrix phase's results are still on their way to the registers
#pragma unroll
            for (int u = 0; u < NSL; ++u) issue_dma();
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int u = 0; u < NSL; ++u) {
                const int i0 = c_sl * BQ;  // first query row of the unit, relative to q_row0
                const bool dead_unit = (NSL > 1) && (c_n >= nunits);
                ++c_n;
                if (++c_hg == p.group) { c_hg = 0; --c_sl; }
                if (!is_k) {
                    const bool need_mask = dead_unit || (kbase_w + 31 > i0 + shift) || (i0 + BQ > q_len) || (kbase_w + 32 > k_rows) ||
                                           (kbase_w < i0 + BQ - 1 + band_low);
                    float pv[16];
#pragma unroll
#if LV_RS_CFOLD
                    for (int i = 0; i < 16; ++i) pv[i] = fast_exp2(x[u][i]);
#else
                    for (int i = 0; i < 16; ++i) pv[i] = fast_exp2(__builtin_fmaf(x[u][i], c, cst[u][i >> 2][i & 3]));
#endif
                    if (need_mask) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            bool dead = (kj > ii + shift) | (kj < ii + band_low) | (ii >= q_len) | (kj >= k_rows) | dead_unit;
                            pv[i] = dead ? 0.f : pv[i];
                            asm volatile("" : "+v"(pv[i]));  // keeps this a branch
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) y[u][i >> 3][i & 7] = (__bf16)pv[i];
                    *(LDS_PTR(bf16x8, p_ptr + u * PBUF)) = y[u][0];
                    *(LDS_PTR(bf16x8, p_ptr + u * PBUF + 1024)) = y[u][1];
                } else {
                    bf16x8 pf[2];
                    pf[0] = *(LDS_PTR(bf16x8, p_ptr + u * PBUF));
                    pf[1] = *(LDS_PTR(bf16x8, p_ptr + u * PBUF + 1024));
#pragma unroll
                    for (int i = 0; i < 16; ++i)
#if LV_RS_CFOLD
                        y[u][i >> 3][i & 7] = (__bf16)((float)pf[i >> 3][i & 7] * x[u][i]);  // (bf16 -> f32 is exact)
#else
                        y[u][i >> 3][i & 7] = (__bf16)((float)pf[i >> 3][i & 7] * (x[u][i] + cst[u][i >> 2][i & 3]));  // (bf16 -> f32 is exact)
#endif
                }
                asm volatile("" : "+v"(y[u][0]), "+v"(y[u][1]));  // Y is formed here, not next to its first use
            }
#if !LV_RS_DMA_FIRST
            // units of iteration it+2; their slots held iteration it-2, whose last reader finished two barriers ago
#pragma unroll
            for (int u = 0; u < NSL; ++u) issue_dma();
#endif
            __builtin_amdgcn_sched_barrier(0);
            LV_STAMP(tb);
            // leave this iteration's requests in flight, everything older has landed
            if (c_wave) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((DEPTH - 1) * (NSL * PPW + 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((DEPTH - 1) * NSL * PPW) : "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ================= MFMA(it): X of the next iteration's units and acc += A_cols^T Y =================
            LV_STAMP(tc);
            __builtin_amdgcn_s_setprio(LV_RS_PRIO_M);
#if LV_RS_CFOLD
#pragma unroll
            for (int u = 0; u < NSL; ++u) load_consts(u, sb_next + u * SLOTB);   // x of the next units starts from their row constants
            load_group(0, 0, sb_next, sb);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NSL + XPG + 2 * APG, 0);
#else
#pragma unroll
            for (int u = 0; u < NSL; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[u][i] = 0.f;
            load_group(0, 0, sb_next, sb);
            __builtin_amdgcn_sched_group_barrier(0x100, XPG + 2 * APG, 0);
#endif
#pragma unroll
            for (int gg = 0; gg < NGT; ++gg) {
                const int set = gg & 1;
                const int u = gg / NG, g = gg % NG;
                if (gg + 1 < NGT) {
                    const int un = (gg + 1) / NG;
                    load_group((gg + 1) % NG, set ^ 1, sb_next + un * SLOTB, sb + un * SLOTB);
                }
#pragma unroll
                for (int i = 0; i < XPG; ++i) {
                    x[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rfr[set][i], bfr[XPG * g + i], x[u], 0, 0, 0);
                    const int j = APG * g + i;
                    acc[j % MT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cfr[set][i], y[u][j / MT], acc[j % MT], 0, 0, 0);
                }
                if (gg + 1 < NGT) {
                    // issue order: the next group's reads (3 per 2 MFMAs) between this group's MFMAs
#pragma unroll
                    for (int i = 0; i < XPG; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * XPG, 0);
                }
            }
#if !LV_RS_CFOLD
#pragma unroll
            for (int u = 0; u < NSL; ++u) load_consts(u, sb_next + u * SLOTB);
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(LV_RS_PRIO_V);
#ifdef LV_RS_STAMP
            asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(x[0][0]));   // the matrix work's results have to exist here
#endif
            LV_STAMP(td);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            LV_STAMP(te);
#ifdef LV_RS_STAMP
            st_vec += tb - ta; st_bar1 += tc - tb; st_mat += td - tc; st_bar2 += te - td;
#endif
            sb = sb_next;
        }
#ifdef LV_RS_STAMP
        if (lane == 0 && (w == 0 || w == 4)) {
            unsigned long long* o = lv_rs_stamps[is_k ? 1 : 0];
            atomicAdd(o + 0, (unsigned long long)st_vec); atomicAdd(o + 1, (unsigned long long)st_bar1);
            atomicAdd(o + 2, (unsigned long long)st_mat); atomicAdd(o + 3, (unsigned long long)st_bar2);
            atomicAdd(o + 4, (unsigned long long)niter); atomicAdd(o + 5, 1ull);
        }
#endif
        if (!is_k) asm volatile("s_barrier" ::: "memory");
    }
    // no staging request may outlive the workgroup: its LDS could already belong to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: accumulators (key on the lane) -> rows of dv (V-wave) or dk (K-wave, times scale)
    if (kj < k_rows) {
        const int64_t grow = k_row0 + dsc.out_delta + kj;
        const float f = is_k ? p.scale : 1.f;
        void* const gbase = is_k ? p.dk : p.dv;
        const int64_t g_rs = is_k ? p.dk_rs : p.dv_rs, g_hs = is_k ? p.dk_hs : p.dv_hs;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * mt + 8 * g4 + 4 * h;
                if constexpr (ACCUM_F32) {
                    float* gp = reinterpret_cast<float*>(gbase) + grow * g_rs + (int64_t)kvh * g_hs + dcol;
                    f32x4 a = p.f32_store ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(gp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += acc[mt][4 * g4 + e] * f;
                    *reinterpret_cast<f32x4*>(gp) = a;
                } else {
                    __bf16* gp = reinterpret_cast<__bf16*>(gbase) + grow * g_rs + (int64_t)kvh * g_hs + dcol;
                    bf16x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (__bf16)(acc[mt][4 * g4 + e] * f);
                    *reinterpret_cast<bf16x4*>(gp) = a;
                }
            }
    }
}

static int g_rs_units = 0;   // 0: environment not read yet
static int rs_units() {      // LVATTN_DKDV_UNITS=1|2: 32-row units per iteration
    if (g_rs_units == 0) {
        const char* e = getenv("LVATTN_DKDV_UNITS");
        g_rs_units = (e && e[0] == '1') ? 1 : 2;
    }
    return g_rs_units;
}
void lv_select_rs_units(int n) { g_rs_units = (n == 1) ? 1 : 2; }

template <int D, int NSL>
static void launch_rs(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    dim3 grid((unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8)), block(512);
    constexpr int depth = (NSL == 1) ? LV_RS_DEPTH1 : 2;
    size_t lds = (size_t)(depth + 2) * NSL * (2 * 32 * D * 2 + 256) + (size_t)NSL * 4 * 2048;  // ring slots + P buffers
    if (causal) {
        if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_rs_kernel<D, true, true, NSL>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dkdv_rs_kernel<D, true, false, NSL>), grid, block, lds, stream, p);
    } else {
        if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_rs_kernel<D, false, true, NSL>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dkdv_rs_kernel<D, false, false, NSL>), grid, block, lds, stream, p);
    }
}

int lv_launch_bwd_dkdv_rs(const BwdParams& p, int d, int causal, int accum, hipStream_t stream) {
    if (p.n_blocks <= 0) return 0;
    const int two = rs_units() == 2;
    if (d == 128) { if (two) launch_rs<128, 2>(p, causal, accum, stream); else launch_rs<128, 1>(p, causal, accum, stream); }
    else { if (two) launch_rs<64, 2>(p, causal, accum, stream); else launch_rs<64, 1>(p, causal, accum, stream); }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

#ifdef LV_RS_STAMP
extern "C" int lv_debug_rs_stamps(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lv_rs_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lv_rs_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
