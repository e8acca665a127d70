// Flash-style attention forward for gfx950, "ping-pong" form: the two waves that share a SIMD
// never want the same pipe at the same time.
//
// One workgroup = 8 waves = one 256-row query block of one query head (wave w: rows 32w .. 32w+31),
// waves 0-3 form group A, waves 4-7 group B; SIMD i hosts wave i and wave 4+i.  Per 64-key tile a
// wave alternates two phases, separated by workgroup barriers:
//   VALU(s)   : softmax of S_s (row max, deferred rescale, exp2, row sum, cast to bf16 P_s), plus the
//               request for a K or V tile two tiles ahead (direct global -> LDS loads)
//   MFMA(s+1) : S_{s+1}^T = K_{s+1} Q^T  and  O^T += V_s^T P_s^T   (32 v_mfma_f32_32x32x16_bf16)
// and group B runs one barrier behind group A, so that on every SIMD one wave is in its matrix phase
// while the other is in its vector phase: the matrix pipe always has a wave feeding it, and the
// softmax issues into the 24 of every 32 cycles that an MFMA leaves free on the SIMD's issue port.
// The product is software-pipelined by one tile (S of the next tile is formed while P of this one
// is consumed), so S and P of a wave never wait on each other inside a phase.
//
// K/V tiles live in a 3-slot LDS ring (slot s%3: K tile s and V tile s-1, the operands of MFMA(s));
// group A requests K tiles, group B V tiles, each wave 1/4 of a tile per VALU phase; a counted
// s_waitcnt vmcnt before the phase-ending barrier leaves the newest request in flight.
#include "internal.h"

#ifndef LV_PP_MAP
#define LV_PP_MAP 2   // workgroup -> (KV head, query block, head) mapping: 0 = plain order, 1 / 2 = XCD-aware (see the kernel)
#endif
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_PP_NS
#define LV_PP_NS 3
#endif
#ifndef LV_PP_PRIO
#define LV_PP_PRIO 2   // s_setprio level of a wave in its matrix phase (0 in the vector phase)
#endif

template <int D>
__device__ __forceinline__ int lds_off_pp(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

// Direct global -> LDS load of 16 bytes per lane: lane i's bytes land at LDS byte address
// lds_dst + 16 i (lds_dst wave-uniform, carried in M0).  Written as asm so that the compiler does
// not count it (it would wait for all of them before the next transposed LDS read); completion is
// waited for by hand.
__device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

template <int D, bool CAUSAL, bool ROPE = false>
__global__ __launch_bounds__(512) void lv_fwd_pp_kernel(FwdParams p) {
    constexpr int BN = 64;            // keys per tile
    constexpr int KS = D / 16;        // k-steps of the QK^T product
    constexpr int MT = D / 32;        // 32-wide d sub-tiles of the PV product
    constexpr int ROWB = D * 2;       // bytes per K/V row
    constexpr int TILEB = BN * ROWB;  // bytes per K or V tile
    constexpr int CPR = D / 8;        // 16-byte chunks per row
    constexpr int NS = LV_PP_NS;      // ring slots: tiles are requested NS - 1 ahead of the tile being consumed
    constexpr int SLOTB = 2 * TILEB;  // [K tile][V tile]
    constexpr int RPP = 64 / CPR;     // rows per 1-KiB piece (one wave-instruction of the staging)
    constexpr int PPW = (TILEB / 1024) / 4;  // pieces per wave per tile: 4 (d=128) or 2 (d=64)
    constexpr float RESCALE_LOG2 = 8.0f;
    constexpr float OPTIMISTIC_MAX = 4096.0f;  // largest half-row sum of P the optimistic pass accepts (every P <= 2^12)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds0 = (lds_char*)smem;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp_b = (w >= 4);
    const int wg = w & 3;
    const int r = lane & 31;
    const int h = lane >> 5;

#if LV_PP_MAP == 0
    const int bid = blockIdx.x;
    const int hg = bid % p.group;
    const int rest = bid / p.group;
    const int blk = rest % p.n_blocks;
    const int kvh = rest / p.n_blocks;
#else
    // Workgroup i runs on XCD i % 8 (each XCD has its own L2).  The query heads of one (KV head, query block)
    // unit take consecutive slots of ONE XCD, so they stream the same K / V tiles through the same L2 side by side;
    // units go round the XCDs in heaviest-first order.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int hg, blk, kvh;
    if (p.head_per_xcd) {   // all work items of a KV head on one XCD, back to back (see lv_head_per_xcd)
        const int per = p.n_blocks * p.group;
        kvh = (slot / per) * 8 + xcd;
        if (kvh >= p.hkv) return;
        const int rem = slot % per;
        blk = rem / p.group;
        hg = rem % p.group;
    } else {
        hg = slot % p.group;
        const int unit = (slot / p.group) * 8 + xcd;
        if (unit >= p.n_blocks * p.hkv) return;
#if LV_PP_MAP == 1
        blk = unit / p.hkv, kvh = unit % p.hkv;      // with 8 KV heads: one KV head per XCD
#else
        blk = unit % p.n_blocks, kvh = unit / p.n_blocks;
#endif
    }
#endif
    const int head = kvh * p.group + hg;

    const AttnDesc dsc = lv_fwd_desc(p, blk, 256, CAUSAL);
    const int q_row0 = dsc.a_row0, q_rows = dsc.a_rows;
    const int k_row0 = dsc.b_row0, k_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;

    // ---- Q fragments: B operand of S^T = K Q^T : lane holds Q[row r][16ks + 8h .. +8]
    const int qi = 32 * w + r;  // row inside the block
    const int q_row = q_row0 + (qi < q_rows ? qi : q_rows - 1);
    const __bf16* qptr = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * h;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qptr + 16 * ks);
    if constexpr (ROPE) {
        // Rotary embedding of the query row on the lane: column c pairs with c + D/2, i.e. fragment ks with ks + KS/2
        // of the same lane.  The cos / sin row comes in with the same addressing as the query row; the op order and
        // the bf16 roundings are those of lv_rope_kernel, so the rotated row is bit-identical to the standalone pass.
        int64_t trow = p.rope_pos ? (int64_t)p.rope_pos[q_row] : (int64_t)q_row / p.rope_tab_div;
        trow = trow < 0 ? 0 : (trow >= p.rope_tab_rows ? p.rope_tab_rows - 1 : trow);
        const __bf16* cp = p.rope_cos + trow * D + 8 * h;
        const __bf16* sp = p.rope_sin + trow * D + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) {
            const bf16x8 c1 = *reinterpret_cast<const bf16x8*>(cp + 16 * ks), c2 = *reinterpret_cast<const bf16x8*>(cp + 16 * ks + D / 2);
            const bf16x8 s1 = *reinterpret_cast<const bf16x8*>(sp + 16 * ks), s2 = *reinterpret_cast<const bf16x8*>(sp + 16 * ks + D / 2);
            const bf16x8 x1 = qf[ks], x2 = qf[ks + KS / 2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r1, r2;
                rope_pair<false>((float)x1[e], (float)x2[e], (float)c1[e], (float)s1[e], (float)c2[e], (float)s2[e], r1, r2);
                qf[ks][e] = as_bf16(rb(r1));
                qf[ks + KS / 2][e] = as_bf16(rb(r2));
            }
        }
        if (p.q_rot && qi < q_rows) {   // the rotated queries, for the backward pass
            __bf16* qr = p.q_rot + (int64_t)q_row * p.qr_rs + (int64_t)head * p.qr_hs + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) *reinterpret_cast<bf16x8*>(qr + 16 * ks) = qf[ks];
        }
    }

    int limit = qi + shift;  // visible keys (relative to k_row0) for this lane's row: j <= limit
    if (limit > k_len - 1 || qi >= q_rows) limit = k_len - 1;
    const int band_low = CAUSAL ? lv_band_low(dsc) : -(1 << 29);   // ... and j >= qi + band_low when the descriptor has a band
    const int low = qi + band_low;
    int wave_kmax = 32 * w + 32 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    const int my_nkt = (32 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging: group A brings K tiles, group B V tiles; piece j of wave wg is rows
    // RPP*(PPW*wg + j) .. +RPP-1 of the tile.  The LDS image is XOR-swizzled inside each row and the
    // LDS side of the transfer is lane-linear, so the swizzle goes on the source address.
    const int64_t x_rs = grp_b ? p.v_rs : p.k_rs;
    const char* xbase = grp_b ? reinterpret_cast<const char*>(p.v + (int64_t)kvh * p.v_hs + (int64_t)k_row0 * p.v_rs)
                              : reinterpret_cast<const char*>(p.k + (int64_t)kvh * p.k_hs + (int64_t)k_row0 * p.k_rs);
    const int64_t x_tile_bytes = (int64_t)BN * x_rs * 2;
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);
    const int rows_last = rows_avail - ((rows_avail + BN - 1) / BN - 1) * BN;  // rows of the last (maybe partial) tile
    const int prow0 = RPP * PPW * wg + lane / CPR;
    int voff_full[PPW], voff_part[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        const int ch16 = lds_off_pp<D>(row, lane % CPR) - row * ROWB;  // position -> chunk: the swizzle is an involution
        voff_full[j] = row * (int)(x_rs * 2) + ch16;
        voff_part[j] = min(row, rows_last - 1) * (int)(x_rs * 2) + ch16;  // rows past the end repeat the last valid one
    }
    int ld_t = 0;  // next tile to request
    // slot of K tile t: t % NS; of V tile t: (t + 1) % NS (its V half)
    uint32_t ld_dst = lds_base + (grp_b ? SLOTB + TILEB : 0) + PPW * wg * 1024;
    const uint32_t ld_wrap = lds_base + NS * SLOTB;
    auto issue_tile = [&]() __attribute__((always_inline)) {
        int rows = rows_avail - ld_t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        // tiles past the end of the key range: zero records, every lane out of range, nothing fetched
        const uint32_t rec = rows ? (uint32_t)((rows - 1) * x_rs * 2 + ROWB) : 0u;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + ld_t * x_tile_bytes), 0, rec, 0x00020000);
        const bool partial = rows < BN;
#pragma unroll
        for (int j = 0; j < PPW; ++j) pp_dma16(rs, partial ? voff_part[j] : voff_full[j], ld_dst + j * 1024);
        ++ld_t;
        ld_dst = (ld_dst + SLOTB >= ld_wrap) ? ld_dst + SLOTB - NS * SLOTB : ld_dst + SLOTB;
    };

    // ---- LDS read addresses in slot 0
    lds_char* k_ptr[KS];  // K row read: row 32*st + r, chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = lds0 + lds_off_pp<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* v_ptr[MT][2];  // V transposed read (see attn_fwd.hip)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            v_ptr[mt][jh] = lds0 + TILEB + lds_off_pp<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);

    f32x16 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[mt][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale_log2e;
    f32x16 s[2];
    bf16x8 pf[2][2];

    // retire the Q loads before any staging request is in flight
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    auto qk = [&](int sb) __attribute__((always_inline)) {  // S^T = K Q^T from the slot at byte offset sb
#pragma unroll
        for (int st = 0; st < 2; ++st) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[st][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf = *(LDS_PTR(bf16x8, k_ptr[ks] + sb + st * 32 * ROWB));
                s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[st], 0, 0, 0);
            }
        }
    };

    if (nkt > 0) {
        // prologue: tiles 0 and 1 of this group's operand, then S_0 (both groups at once)
#pragma unroll
        for (int i = 0; i < NS - 1; ++i) issue_tile();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        qk(0);
        int sb = 0;  // byte offset of the slot of the NEXT matrix phase
        if (grp_b) asm volatile("s_barrier" ::: "memory");
        for (int t = 0; t < nkt; ++t) {
            // ================= VALU(t): softmax of S_t -> P_t =================
            if (t < my_nkt) {
                const int kb0 = t * BN;
                const bool need_mask = (kb0 + BN - 1 > 32 * w + shift) || (kb0 + BN > k_len) || (kb0 < 32 * w + 31 + band_low);
                // Optimistic pass: once every row of the wave has a finite running maximum, P is formed
                // against it WITHOUT looking for the tile's maximum first (a third of the softmax's
                // instructions).  Scores far below the running maximum just give tiny P; a score far above
                // it shows up as a large row sum, and only then the exact pass below redoes the tile.
                bool exact = need_mask || __any(m_run < -1e29f);
                if (!exact) {
                    const float mc = m_run * c;
                    float rs = 0.f;
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float pv = fast_exp2(__builtin_fmaf(s[st][i], c, -mc));
                            rs += pv;
                            pf[st][i >> 3][i & 7] = (__bf16)pv;
                        }
                    if (__any(!(rs <= OPTIMISTIC_MAX))) exact = true;   // (also catches inf / nan)
                    else l_run += rs;
                }
                if (exact) {
                if (need_mask) {
#pragma unroll
                    for (int st = 0; st < 2; ++st)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            int j = kb0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h;
                            s[st][i] = (j > limit || j < low) ? -INFINITY : s[st][i];
                        }
                }
                float mx = s[0][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
                for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
                mx = pair32_max(mx);
                // Deferred rescale: the running maximum is raised (and O, l rescaled) only when some row
                // of the wave saw its maximum grow by more than 2^RESCALE_LOG2.
                const float m_new = fmaxf(m_run, mx);
                const bool grow = (m_new - m_run) * c > RESCALE_LOG2;
                if (__any(grow)) {
                    const float alpha = fast_exp2((m_run - m_new) * c);
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) oacc[mt][i] *= alpha;
                }
                const float mc = m_run * c;
                float rs = 0.f;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float pv = fast_exp2(__builtin_fmaf(s[st][i], c, -mc));
                        rs += pv;
                        pf[st][i >> 3][i & 7] = (__bf16)pv;
                    }
                l_run += rs;
                }
            } else {
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int i = 0; i < 8; ++i) pf[st][s2][i] = (__bf16)0.f;
            }
            issue_tile();  // K (group A) or V (group B) tile t + NS - 1
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * PPW) : "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ================= MFMA(t+1): S_{t+1} = K_{t+1} Q^T, O += V_t P_t =================
            // The wave in its matrix phase outranks its SIMD partner (in its vector phase) at the issue
            // port: an MFMA goes out the moment the pipe is free, the softmax fills the cycles between.
            __builtin_amdgcn_s_setprio(LV_PP_PRIO);
            sb = (sb + SLOTB == NS * SLOTB) ? 0 : sb + SLOTB;
            // Four groups of 8 MFMAs: group (st, s2) = MT products O += V[32st+16s2 .. +16]^T P[st][s2] and
            // KS/2 steps of the S[st] chain.  The fragments of group g+1 are read from LDS while group g
            // runs (two register sets), so that no MFMA waits for a read issued just before it.
            bf16x8 vfr[2][MT], kfr[2][KS / 2];
            auto load_group = [&](int g, int set) __attribute__((always_inline)) {
                const int st = g >> 1, s2 = g & 1;
                const int voff = sb + (32 * st + 16 * s2) * ROWB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][0] + voff));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][1] + voff));
                    bf16x8& vf = vfr[set][mt];
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                }
#pragma unroll
                for (int kk = 0; kk < KS / 2; ++kk)
                    kfr[set][kk] = *(LDS_PTR(bf16x8, k_ptr[s2 * (KS / 2) + kk] + sb + st * 32 * ROWB));
            };
            constexpr int RPG = 2 * MT + KS / 2;  // LDS reads per group
            constexpr int MPG = MT + KS / 2;      // MFMAs per group
            load_group(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, RPG, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int st = g >> 1, s2 = g & 1, set = g & 1;
                if (g < 3) load_group(g + 1, set ^ 1);
                if (s2 == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) s[st][i] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < MPG; ++i) {
                    // alternate the two accumulation targets so that consecutive MFMAs are independent
                    if (i % 2 == 0 && i / 2 < MT) oacc[i / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[set][i / 2], pf[st][s2], oacc[i / 2], 0, 0, 0);
                    else if (i % 2 == 1 && i / 2 < KS / 2) s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[set][i / 2], qf[s2 * (KS / 2) + i / 2], s[st], 0, 0, 0);
                    else if (i / 2 >= KS / 2) oacc[i - KS / 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[set][i - KS / 2], pf[st][s2], oacc[i - KS / 2], 0, 0, 0);
                    else s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[set][i - MT], qf[s2 * (KS / 2) + i - MT], s[st], 0, 0, 0);
                }
                // issue order: the next group's reads spread between this group's MFMAs
                if (g < 3) {
                    static_assert(RPG * 2 == MPG * 3, "three reads per two MFMAs");
#pragma unroll
                    for (int i = 0; i < MPG / 2; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!grp_b) asm volatile("s_barrier" ::: "memory");
        // no staging request may outlive the workgroup: its LDS could already belong to another one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: normalise, store O (row-per-lane, 8-byte pieces) and LSE
    const float l_tot = pair32_sum(l_run);
    const bool live = (qi < q_rows);
    const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
    if (live) {
        __bf16* optr = p.o + (int64_t)(q_row0 + dsc.out_delta + qi) * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (__bf16)(oacc[mt][4 * g4 + e] * inv);
                *reinterpret_cast<bf16x4*>(optr + 32 * mt + 8 * g4 + 4 * h) = ov;
            }
        if (h == 0) {
            float lse = (l_tot > 0.f) ? (m_run * p.scale + __logf(l_tot)) : -INFINITY;
            p.lse[(int64_t)head * p.lse_stride + q_row0 + dsc.out_delta + qi] = lse;
        }
    }
}

template <int D>
static void launch_pp(const FwdParams& p0, int causal, hipStream_t stream) {
    FwdParams p = p0;
#if LV_PP_MAP == 0
    dim3 grid((unsigned)((int64_t)p.n_blocks * p.hq)), block(512);
#else
    p.head_per_xcd = lv_head_per_xcd(p) ? 1 : 0;
    dim3 grid(p.head_per_xcd ? (unsigned)(((p.hkv + 7) / 8) * 8 * p.n_blocks * p.group)
                             : (unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8 * p.group)), block(512);
#endif
    size_t lds = (size_t)LV_PP_NS * 2 * 64 * D * 2;  // ring slots x (K tile + V tile)
    if constexpr (D == 128) {
        if (p.rope_cos) {
            if (causal) hipLaunchKernelGGL((lv_fwd_pp_kernel<D, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_fwd_pp_kernel<D, false, true>), grid, block, lds, stream, p);
            return;
        }
    }
    if (causal) hipLaunchKernelGGL((lv_fwd_pp_kernel<D, true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_fwd_pp_kernel<D, false>), grid, block, lds, stream, p);
}

int lv_launch_fwd_pp(const FwdParams& p, int d, int causal, hipStream_t stream) {
    if (p.rope_cos && d != 128) return -3;   // the fused rotation exists for head dim 128 only
    if (d == 128) launch_pp<128>(p, causal, stream);
    else launch_pp<64>(p, causal, stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
