// Flash-style attention forward for gfx950, "one wave per SIMD" form (head dim 128; the default when a work item sweeps at least 32,768 keys,
// LVATTN_FWD_KERNEL=a4 forces it).
//
// One workgroup = 4 waves = one 256-row query block of one query head.  A wave owns 64 query rows (two 32-row column
// blocks c = 0, 1 of S^T = K Q^T, query row on the lane) and the WHOLE 512-register file of its SIMD, laid out by hand:
//   accumulator half   a[0:127]    O^T, tile (c, mt) at a[64 c + 16 mt .. +15]
//                      a[128:191]  Q fragments, (c, ks) at a[128 + 32 c + 4 ks .. +3]
//                      a[192:255]  the K tile's fragments, (st, ks) at a[192 + 32 st + 4 ks .. +3]
//   vector half        scores, P, V fragments, addresses (allocated by the compiler)
// The MFMAs are inline asm naming those accumulator registers, so every K / V fragment read from LDS feeds TWO MFMAs
// (one per column block): half the LDS read bytes per MFMA of the 8-wave kernels, and no partner wave or second
// barrier per tile.  The compiler neither counts nor pads what is inside an asm statement: the waits and the wait
// states (a vector write -> MFMA operand: 2; an MFMA result -> any other reader: 12) are written out here.
// The build must show no compiler v_accvgpr_* and no spill (the vector half has to stay within 256 registers): the
// Makefile runs audit_a4.py over this file's assembly; see the top of the kernel for how the accumulator half is kept
// out of the register allocator's hands.
//
// The loop is software-pipelined: iteration t holds O^T += V_{t-1}^T P_{t-1}^T, S_{t+1}^T = K_{t+1} Q^T and the softmax of
// S_t, the vector work spread over the 64 MFMA gaps of the iteration; one workgroup barrier per 64-key tile.
// K and V tiles live in 4-slot LDS rings; iteration t requests K tile t+4 and V tile t+1 by direct global -> LDS loads
// (each wave a quarter of each), a counted s_waitcnt vmcnt before the barrier leaves the newest iteration's requests in
// flight.  K runs further ahead because its fragments move into the accumulator registers one iteration before the S
// product that reads them.
//
// One wave per SIMD issues every instruction of its SIMD, one per four cycles at best whatever the kind, so the loop is
// kept short: it is unrolled four times, which makes every ring slot a compile-time constant (LDS addresses are one lane
// register plus an immediate, staging destinations one scalar add into M0), the staging descriptors are advanced
// instead of rebuilt, and the per-tile decisions are one scalar compare against a precomputed tile index.
#include "a4_asm.h"

// accumulator register numbers
__device__ constexpr int a4_O(int c, int mt) { return 64 * c + 16 * mt; }
__device__ constexpr int a4_Q(int c, int ks) { return 128 + 32 * c + 4 * ks; }
__device__ constexpr int a4_K(int st, int ks) { return 192 + 32 * st + 4 * ks; }

// LDS reads of one iteration, by gap (each gap: its MFMA first, then its V reads, then its K read).  All sit in even
// gaps; the odd ones are full with the second half of the softmax share.
//   K_{t+2} fragment f = (st, ks) in gap 6 + 2 f: the S product is done with that register two gaps earlier at the latest
//   V_{t-1} fragment (g, mt): group 0 in gaps 24 .. 30, group g >= 1 in the gap after the MFMAs of fragment (g - 1, mt),
//   whose registers it reuses, six gaps ahead of its own MFMAs
__device__ constexpr int a4_nk(int k) { return (!(k & 1) && k >= 6 && k <= 36) ? 1 : 0; }
__device__ constexpr int a4_nv(int k) { return (!(k & 1) && ((k >= 24 && k <= 30) || (k >= 34 && k <= 56))) ? 2 : 0; }
__device__ constexpr int a4_lds_before(int k) { int n = 0; for (int i = 0; i < k; ++i) n += a4_nk(i) + a4_nv(i); return n; }
__device__ constexpr int a4_v_gap(int g, int mt) { return g == 0 ? 24 + 2 * mt : 32 + 8 * (g - 1) + 2 * mt + 2; }
// reads allowed in flight when the MFMA of gap k (PV half, group g) needs the fragments up to (g, mt_last)
__device__ constexpr int a4_v_pending(int k, int mt_last) {
    return a4_lds_before(k) - (a4_lds_before(a4_v_gap((k - 32) >> 3, mt_last)) + 2);
}
// the staging requests of an iteration: K pieces 0-3 in gaps 3, 7, 11, 15 and V pieces 0-3 in gaps 19, 23, 27, 31; the V
// descriptor moves on in gap 1, the K descriptor in gap 17.  (Spread over the first half of the iteration like this the
// workgroups of an XCD keep sweeping K / V together, L2 hit rate 95 % at 1M tokens; bunched behind the barrier, or with V
// requested two tiles ahead, 65-73 %, seven times the memory-side traffic and 1.3 % more time.)
__device__ constexpr int a4_dma_piece(int k) {   // 0-3: K piece, 4-7: V piece, -1: none
    return (k < 32 && (k & 3) == 3) ? (k >> 2) : -1;
}
template <bool CAUSAL>
__global__ __launch_bounds__(256) void lv_fwd_a4_kernel(FwdParams p) {
    constexpr int D = 128;
    constexpr int BN = 64;
    constexpr int KS = D / 16;        // 8
    constexpr int MT = D / 32;        // 4
    constexpr int ROWB = D * 2;
    constexpr int TILEB = BN * ROWB;  // 16 KiB
    constexpr int CPR = D / 8;
    constexpr int NS = 4;             // ring slots: K tiles t+1 .. t+4, V tiles t-1 .. t+1 around iteration t (a power of two: slot = t mod 4)
    constexpr int KRING = NS * TILEB; // byte offset of the K ring (the V ring comes first: every read offset fits an immediate)
    constexpr int RPP = 64 / CPR;     // rows per 1-KiB piece: 4
    constexpr int PPW = 4;            // pieces per wave per tile and operand
    constexpr float RESCALE_LOG2 = 8.0f;
    constexpr float OPTIMISTIC_MAX = 4096.0f;

    // The accumulator half belongs to the asm statements below.  Sixteen 16-register values of the accumulator class are
    // declared here and kept alive to the end of the kernel: all 256 registers then count as occupied for the register
    // allocator, which would otherwise park its own values in them (it does, as soon as the vector half is full).  Which
    // value it maps to which register does not matter: it never touches them, only the asm statements do, by number.
    f32x16 hold[16];
    asm volatile("" : "=a"(hold[0]), "=a"(hold[1]), "=a"(hold[2]), "=a"(hold[3]), "=a"(hold[4]), "=a"(hold[5]), "=a"(hold[6]), "=a"(hold[7]),
                      "=a"(hold[8]), "=a"(hold[9]), "=a"(hold[10]), "=a"(hold[11]), "=a"(hold[12]), "=a"(hold[13]), "=a"(hold[14]), "=a"(hold[15]));

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds0 = (lds_char*)smem;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // workgroup -> (KV head, query block, head in group): as in the ping-pong kernel
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int hg, blk, kvh;
    if (p.head_per_xcd) {
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
        blk = unit % p.n_blocks, kvh = unit / p.n_blocks;
    }
    const int head = kvh * p.group + hg;

    const AttnDesc dsc = lv_fwd_desc(p, blk, 256, CAUSAL);
    // (wave-uniform by construction; said so explicitly, or the merged dense / descriptor value sits in a vector register)
    const int q_row0 = __builtin_amdgcn_readfirstlane(dsc.a_row0), q_rows = __builtin_amdgcn_readfirstlane(dsc.a_rows);
    const int k_row0 = __builtin_amdgcn_readfirstlane(dsc.b_row0), k_len = __builtin_amdgcn_readfirstlane(dsc.b_len);
    const int shift = CAUSAL ? __builtin_amdgcn_readfirstlane(dsc.shift) : 0x3fffffff;
    const int o_row0 = q_row0 + __builtin_amdgcn_readfirstlane(dsc.out_delta);   // where the results go (workspace rows for split sweeps)

    // ---- Q fragments of the two column blocks into a[128:191]: lane holds Q[row][16ks + 8h .. +8]
    // row of column block c inside the work item, and the last key it sees (recomputed where needed: registers are short)
    auto row_of = [&](int c) { return 64 * w + 32 * c + r; };
    {
        // one column block at a time: sixteen fragments in flight together would not fit beside everything else
        a4_u32x4 qv[KS];
        auto load_q = [&](int c) __attribute__((always_inline)) {
            const int q_row = q_row0 + (row_of(c) < q_rows ? row_of(c) : q_rows - 1);
            const __bf16* qptr = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qv[ks] = *reinterpret_cast<const a4_u32x4*>(qptr + 16 * ks);
        };
#define A4_PUTQ(c, ks) a4_put4<a4_Q(c, ks)>(qv[ks]);
        load_q(0);
        A4_PUTQ(0, 0) A4_PUTQ(0, 1) A4_PUTQ(0, 2) A4_PUTQ(0, 3) A4_PUTQ(0, 4) A4_PUTQ(0, 5) A4_PUTQ(0, 6) A4_PUTQ(0, 7)
        __builtin_amdgcn_sched_barrier(0);
        load_q(1);
        A4_PUTQ(1, 0) A4_PUTQ(1, 1) A4_PUTQ(1, 2) A4_PUTQ(1, 3) A4_PUTQ(1, 4) A4_PUTQ(1, 5) A4_PUTQ(1, 6) A4_PUTQ(1, 7)
        __builtin_amdgcn_sched_barrier(0);
#undef A4_PUTQ
    }
    // O^T = 0
    a4_zero16<0>(); a4_zero16<16>(); a4_zero16<32>(); a4_zero16<48>(); a4_zero16<64>(); a4_zero16<80>(); a4_zero16<96>(); a4_zero16<112>();

    int wave_kmax = 64 * w + 64 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    const int my_nkt = (64 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging: wave w brings rows 16w .. 16w+15 of every K tile and of every V tile (4 pieces each)
    const char* kbase = reinterpret_cast<const char*>(p.k + (int64_t)kvh * p.k_hs + (int64_t)k_row0 * p.k_rs);
    const char* vbase = reinterpret_cast<const char*>(p.v + (int64_t)kvh * p.v_hs + (int64_t)k_row0 * p.v_rs);
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);
    const int prow0 = RPP * PPW * w + lane / CPR;
    // Piece j of a tile is rows prow0 + RPP j .. of it.  The lane's offset in piece 0 is its row times the row stride
    // plus its swizzled position in the row; since prow0 / 4 is a multiple of 4, piece j differs by j in the low bits
    // of the position (an XOR: row strides are multiples of 256 bytes here, the launcher sees to that) plus j row steps.
    // Rows past the end of the last, partial tile are beyond the descriptor's size and arrive as zeros (masked keys).
    const int ch0 = lds_off128(prow0, lane % CPR) - prow0 * ROWB;
    const int kvo0 = prow0 * (int)(p.k_rs * 2) + ch0, vvo0 = prow0 * (int)(p.v_rs * 2) + ch0;
    const int k_row_bytes = __builtin_amdgcn_readfirstlane((int)(p.k_rs * 2)), v_row_bytes = __builtin_amdgcn_readfirstlane((int)(p.v_rs * 2));
    const int kstep = RPP * k_row_bytes, vstep = RPP * v_row_bytes;
    const int kstep2 = 2 * kstep, kstep3 = 3 * kstep, vstep2 = 2 * vstep, vstep3 = 3 * vstep;
    const uint32_t wave_base = lds_base + (uint32_t)(PPW * w * 1024);
    // A tile's descriptor: base = its first row, size = its valid rows x the row stride (a row's 256 bytes end before the
    // next row starts, so exactly the valid rows pass the range check).  Kept as four scalars and moved on tile by tile.
    struct Stage { a4_u32x4 rs; int rem; };
    auto stage_at = [&](const char* base, int64_t row_elems, int t) __attribute__((always_inline)) {
        const uint64_t b = (uint64_t)(uintptr_t)(base + (int64_t)t * BN * row_elems * 2);
        Stage st;
        st.rem = rows_avail - t * BN;
        const int rows = st.rem < 0 ? 0 : (st.rem > BN ? BN : st.rem);
        st.rs[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
        st.rs[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) & 0xffffu;
        st.rs[2] = (uint32_t)__builtin_amdgcn_readfirstlane(rows * (int)(row_elems * 2));
        st.rs[3] = 0x00020000u;
        return st;
    };
    auto stage_next = [&](Stage& st, int row_bytes) __attribute__((always_inline)) {
        const uint64_t b = (((uint64_t)st.rs[1] << 32) | st.rs[0]) + (uint64_t)(uint32_t)(BN * row_bytes);
        st.rs[0] = (uint32_t)b;
        st.rs[1] = (uint32_t)(b >> 32);
        st.rem -= BN;
        int rows = st.rem < 0 ? 0 : st.rem;
        asm volatile("" : "+s"(rows));            // (two scalar steps: written as one clamp it becomes a vector med3)
        rows = rows > BN ? BN : rows;
        st.rs[2] = (uint32_t)(rows * row_bytes);
    };
    // piece j of K tile -> K slot, of V tile -> V slot (prologue form: slot and piece are compile-time there too)
#define A4_DMA_K(st_, slot, j) a4_dma<KRING + (slot) * TILEB + (j) * 1024, j, (j) << 4>((st_).rs, kvo0, (j) == 1 ? kstep : (j) == 2 ? kstep2 : kstep3, wave_base)
#define A4_DMA_V(st_, slot, j) a4_dma<(slot) * TILEB + (j) * 1024, j, (j) << 4>((st_).rs, vvo0, (j) == 1 ? vstep : (j) == 2 ? vstep2 : vstep3, wave_base)

    // ---- LDS read addresses in slot 0 of each ring
    // The XOR swizzle acts on address bits 4..7 only and every slot starts on a multiple of 256 bytes, so the address of
    // fragment ks (chunk 2 ks + h) is the address of fragment 0 with ks XORed into bits 5..7: one register instead of eight.
    const uint32_t k_addr0 = lds_base + (uint32_t)(KRING + lds_off128(r, h));
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    // likewise the d sub-tile mt (chunk 4 mt + ...) of a V read is bits 6..7
    uint32_t v_addr0[2];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh)
        v_addr0[jh] = lds_base + (uint32_t)(lds_off128(8 * jh + 4 * h + q4, 2 * gl + (p4 >> 1)) + 8 * (p4 & 1));

    // running maximum in the exponent's units (score x scale x log2 e) and running sum, per column block
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    const float cs = p.scale_log2e;
    f32x16 sA[2][2], sB[2][2];        // S^T of two consecutive tiles: [column block][32-key half]
    uint32_t pA[2][2][2][4], pB[2][2][2][4];  // P^T (bf16 pairs) of two consecutive tiles: [column block][half][16-key quarter]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pA[c][st][s2][i] = 0u; pB[c][st][s2][i] = 0u; }

    // exact softmax of tile t held in sc (mask, row maximum, deferred rescale) -> pf; updates m_run / l_run / O
    auto softmax_exact = [&](f32x16 (&sc)[2][2], uint32_t (&pf)[2][2][2][4], int t, bool need_mask) __attribute__((always_inline)) {
        const int kb0 = t * BN;
        // (a rare path: the lane's row and column half come from a lane id read here, not from registers kept for it)
        int lane_x;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_x));
        const int r_x = lane_x & 31, h_x = lane_x >> 5;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (need_mask) {
                const int qi_ = 64 * w + 32 * c + r_x;
                const int lim_ = qi_ + shift;
                const int lim = (lim_ > k_len - 1 || qi_ >= q_rows) ? k_len - 1 : lim_;
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int j = kb0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h_x;
                        sc[c][st][i] = (j > lim) ? -INFINITY : sc[c][st][i];
                    }
            }
            float mx = sc[c][0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc[c][0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[c][1][i]);
            mx = pair32_max(mx) * cs;
            const float m_new = fmaxf(m_run[c], mx);
            const bool grow = (m_new - m_run[c]) > RESCALE_LOG2;
            if (__any(grow)) {
                const float alpha = fast_exp2(m_run[c] - m_new);
                m_run[c] = m_new;
                l_run[c] *= alpha;
                if (c == 0) a4_scale<0, 64>(alpha); else a4_scale<64, 64>(alpha);
            }
            const float mc = m_run[c];
            float rs = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float pa = fast_exp2(__builtin_fmaf(sc[c][st][i], cs, -mc));
                    const float pb = fast_exp2(__builtin_fmaf(sc[c][st][i + 1], cs, -mc));
                    rs += pa + pb;
                    a4_pack_into(pf[c][st][i >> 3][(i & 7) >> 1], pa, pb);
                }
            l_run[c] += rs;
        }
    };

    // Tiles below t_fast need no mask for any row of this wave and are live for it: the optimistic pass alone handles them
    // (once every row has seen a key: all_seen).  Everything else, the diagonal tiles, the last partial tile, tiles past
    // the wave's rows, goes through the decisions at the end of the iteration.
    int t_fast = min(my_nkt, min((64 * w + shift + 1) >> 6, k_len >> 6));
    t_fast = t_fast < 0 ? 0 : t_fast;
    int t_fast_now = 0;               // t_fast once all_seen, 0 before

    Stage kst = stage_at(kbase, p.k_rs, 4), vst = stage_at(vbase, p.v_rs, 0);   // K tile 4 for iteration 0; V moves on before its use

    // One iteration (one barrier): 64 MFMAs, each followed by its share of the other work ("gap" k = 0 .. 63):
    //   gaps  0-31  S_{t+1}^T = K_{t+1} Q^T       segment (st, kh), then (ks, c); the K fragments are in a[192:255] already
    //   gaps 32-63  O^T += V_{t-1}^T P_{t-1}^T   group g = (st, s2), then (mt, c)
    //   every gap   one score of S_t per lane: exp2(s x scale - running maximum) (optimistic pass; a tile that needs
    //               masking or whose sums come out too large is redone exactly after the gaps, S_t is still intact)
    //   odd gaps    the pair of exponentials finished two gaps earlier: row sum, cast to bf16 -> P_t
    //   even gaps   LDS reads (V_{t-1} fragments, K_{t+2} fragments -> a[192:255]), on the schedule of a4_nk / a4_nv
    //   8 odd gaps  the staging requests (K tile t+4, V tile t+1), a4_dma_piece
    // SLOT = t mod 4 fixes every ring slot at compile time.
    auto iteration = [&](auto SLOT_, int t) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOT_)::value;
        f32x16 (&scur)[2][2] = (SLOT & 1) ? sB : sA;
        f32x16 (&snext)[2][2] = (SLOT & 1) ? sA : sB;
        uint32_t (&pprev)[2][2][2][4] = (SLOT & 1) ? pA : pB;
        uint32_t (&pcur)[2][2][2][4] = (SLOT & 1) ? pB : pA;
        // everything older than the previous iteration's requests has landed (K tile t+2, V tile t-1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        constexpr int KOFF = ((SLOT + 2) & 3) * TILEB;            // slot of K tile t+2
        constexpr int VOFF = ((SLOT + 3) & 3) * TILEB;            // slot of V tile t-1 (t = 0: zeros, P is zero as well)
        float rs0 = 0.f, rs1 = 0.f;
        float pv[4];              // exponentials in flight: score k in pv[k & 3]
        bf16x4 vh[MT][2];         // V fragments of ONE group, as the two transposed 8-byte reads of each: a fragment's
                                  // registers are re-read for the next group as soon as its two MFMAs have been issued
        // row sum and cast of the pair of scores (E, E + 1)
        auto finish = [&](auto E_) __attribute__((always_inline)) {
            constexpr int e = decltype(E_)::value, c = e >> 5, st = (e >> 4) & 1, i0 = e & 15;
            // (the empty asm statements keep the sums and the cast in THIS gap: left alone, the compiler gathers them at
            // the end of the iteration and keeps all 64 exponentials alive until then)
            if constexpr (c) {
                if constexpr (e == 32) rs1 = pv[e & 3] + pv[(e + 1) & 3]; else { rs1 += pv[e & 3]; rs1 += pv[(e + 1) & 3]; }
                asm volatile("" : "+v"(rs1));
            } else {
                if constexpr (e == 0) rs0 = pv[e & 3] + pv[(e + 1) & 3]; else { rs0 += pv[e & 3]; rs0 += pv[(e + 1) & 3]; }
                asm volatile("" : "+v"(rs0));
            }
            uint32_t pw = a4_pack(pv[e & 3], pv[(e + 1) & 3]);
            asm volatile("" : "+v"(pw));
            pcur[c][st][i0 >> 3][(i0 & 7) >> 1] = pw;
        };
        a4_static_for<0, 64>([&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;
            // ---- the MFMA of this gap
            if constexpr (k < 32) {
                constexpr int sg = k >> 3, st = sg >> 1, kh = sg & 1, ks = 4 * kh + ((k & 7) >> 1), c = k & 1;
                a4_mfma_s<a4_K(st, ks), a4_Q(c, ks), ks == 0>(snext[c][st]);
            } else {
                constexpr int g = (k - 32) >> 3, st = g >> 1, s2 = g & 1, mt = (k & 7) >> 1, c = k & 1;
                bf16x8 vf;
                const bf16x4 lo = vh[mt][0], hi = vh[mt][1];
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                // (fragments mt and mt + 1 are waited for together: half the waits)
                constexpr int wait = (c == 0 && (mt & 1) == 0) ? a4_v_pending(k, mt + 1) : -1;
                const a4_u32x4 pq = {pprev[c][st][s2][0], pprev[c][st][s2][1], pprev[c][st][s2][2], pprev[c][st][s2][3]};
                a4_mfma_o<a4_O(c, mt), wait>(vf, __builtin_bit_cast(bf16x8, pq));
            }
            __builtin_amdgcn_sched_barrier(0);      // the MFMA first, its gap's fillers behind it
            // ---- V fragments
            if constexpr (a4_nv(k) != 0) {
                constexpr int g = k < 32 ? 0 : (k - 34) / 8 + 1, mt = k < 32 ? (k - 24) >> 1 : ((k - 34) & 7) >> 1;
                static_assert(a4_v_gap(g, mt) == k, "V read schedule");
                constexpr int voff = VOFF + (32 * (g >> 1) + 16 * (g & 1)) * ROWB;
                a4_read_v<voff>(vh[mt][0], v_addr0[0] ^ (uint32_t)(mt << 6));
                a4_read_v<voff>(vh[mt][1], v_addr0[1] ^ (uint32_t)(mt << 6));
            }
            // ---- K_{t+2} fragments
            if constexpr (a4_nk(k) != 0) {
                constexpr int f = (k - 6) >> 1, st = f >> 3, ks = f & 7;
                a4_read_k<a4_K(st, ks), KOFF + st * 32 * ROWB>(k_addr0 ^ (uint32_t)(ks << 5));
            }
            // ---- staging requests: K tile t+4 into the slot of K tile t, V tile t+1 into the slot of V tile t-3
            if constexpr (a4_dma_piece(k) >= 0) {
                constexpr int pc = a4_dma_piece(k);
                if constexpr (pc == 0) A4_DMA_K(kst, SLOT, 0);
                else if constexpr (pc == 1) A4_DMA_K(kst, SLOT, 1);
                else if constexpr (pc == 2) A4_DMA_K(kst, SLOT, 2);
                else if constexpr (pc == 3) A4_DMA_K(kst, SLOT, 3);
                else if constexpr (pc == 4) A4_DMA_V(vst, (SLOT + 1) & 3, 0);
                else if constexpr (pc == 5) A4_DMA_V(vst, (SLOT + 1) & 3, 1);
                else if constexpr (pc == 6) A4_DMA_V(vst, (SLOT + 1) & 3, 2);
                else A4_DMA_V(vst, (SLOT + 1) & 3, 3);
            }
            if constexpr (k == 17) stage_next(kst, k_row_bytes);     // -> K tile t+5, for the next iteration
            if constexpr (k == 1) stage_next(vst, v_row_bytes);       // -> V tile t+1, for this one
            // ---- softmax share: score k of the 64 per lane
            {
                constexpr int c = k >> 5, st = (k >> 4) & 1, i = k & 15;
                pv[k & 3] = fast_exp2(__builtin_fmaf(scur[c][st][i], cs, c ? -m_run[1] : -m_run[0]));
                if constexpr ((k & 1) == 1 && k >= 3) finish(std::integral_constant<int, k - 3>{});
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        finish(std::integral_constant<int, 62>{});
        // (no LDS read is outstanding: the last MFMAs waited for the last V reads, and the K reads came before those)
        bool exact;
        if (__builtin_expect(t < t_fast_now, 1)) {
            float rsm;
            asm("v_max_f32 %0, %1, %2" : "=v"(rsm) : "v"(rs0), "v"(rs1));
            exact = __builtin_expect(__any(!(rsm <= OPTIMISTIC_MAX)), 0);
            if (!exact) { l_run[0] += rs0; l_run[1] += rs1; }
        } else {
            exact = (t < my_nkt);
        }
        if (__builtin_expect(exact, 0)) {
            const int kb0 = t * BN;
            softmax_exact(scur, pcur, t, (kb0 + BN - 1 > 64 * w + shift) || (kb0 + BN > k_len));
            t_fast_now = __any(m_run[0] < -1e29f || m_run[1] < -1e29f) ? 0 : t_fast;
        }
        if (__builtin_expect(t >= my_nkt, 0)) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, 0" : "+v"(pcur[c][st][s2][i]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nkt > 0) {
        // prologue: K tiles 0 .. 3 and V tile 0; S_0 from K tile 0, then the fragments of K tile 1 for iteration 0
        {
            Stage k0 = stage_at(kbase, p.k_rs, 0), k1 = stage_at(kbase, p.k_rs, 1), k2 = stage_at(kbase, p.k_rs, 2), k3 = stage_at(kbase, p.k_rs, 3);
            Stage v0 = stage_at(vbase, p.v_rs, 0);
            A4_DMA_K(k0, 0, 0); A4_DMA_K(k0, 0, 1); A4_DMA_K(k0, 0, 2); A4_DMA_K(k0, 0, 3);
            A4_DMA_V(v0, 0, 0); A4_DMA_V(v0, 0, 1); A4_DMA_V(v0, 0, 2); A4_DMA_V(v0, 0, 3);
            A4_DMA_K(k1, 1, 0); A4_DMA_K(k1, 1, 1); A4_DMA_K(k1, 1, 2); A4_DMA_K(k1, 1, 3);
            A4_DMA_K(k2, 2, 0); A4_DMA_K(k2, 2, 1); A4_DMA_K(k2, 2, 2); A4_DMA_K(k2, 2, 3);
            A4_DMA_K(k3, 3, 0); A4_DMA_K(k3, 3, 1); A4_DMA_K(k3, 3, 2); A4_DMA_K(k3, 3, 3);
        }
        // Iteration 0 multiplies P = 0 by the V slot of "tile -1", which nothing has written: zero it (0 x NaN is NaN).
        {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                *(LDS_PTR(f32x4, lds0 + (NS - 1) * TILEB + (PPW * w + j) * 1024 + lane * 16)) = z;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define A4_RK(st, ks, off) a4_read_k<a4_K(st, ks), st * 32 * ROWB + off>(k_addr0 ^ (uint32_t)(ks << 5));
#define A4_RK16(off) A4_RK(0, 0, off) A4_RK(0, 1, off) A4_RK(0, 2, off) A4_RK(0, 3, off) A4_RK(0, 4, off) A4_RK(0, 5, off) A4_RK(0, 6, off) \
                     A4_RK(0, 7, off) A4_RK(1, 0, off) A4_RK(1, 1, off) A4_RK(1, 2, off) A4_RK(1, 3, off) A4_RK(1, 4, off) A4_RK(1, 5, off) \
                     A4_RK(1, 6, off) A4_RK(1, 7, off)
        A4_RK16(0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define A4_S(c, st, ks, first) a4_mfma_s<a4_K(st, ks), a4_Q(c, ks), first>(sA[c][st]);
#define A4_S8(c, st) A4_S(c, st, 0, true) A4_S(c, st, 1, false) A4_S(c, st, 2, false) A4_S(c, st, 3, false) \
                     A4_S(c, st, 4, false) A4_S(c, st, 5, false) A4_S(c, st, 6, false) A4_S(c, st, 7, false)
        A4_S8(0, 0) A4_S8(1, 0) A4_S8(0, 1) A4_S8(1, 1)
#undef A4_S8
#undef A4_S
        a4_nop<12>();
        A4_RK16(TILEB)
#undef A4_RK16
#undef A4_RK
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // iteration t: PV of tile t-1, S of tile t+1, softmax of tile t; one more iteration for the last PV
        for (int t = 0;; t += 4) {
            iteration(std::integral_constant<int, 0>{}, t);
            if (t + 1 > nkt) break;
            iteration(std::integral_constant<int, 1>{}, t + 1);
            if (t + 2 > nkt) break;
            iteration(std::integral_constant<int, 2>{}, t + 2);
            if (t + 3 > nkt) break;
            iteration(std::integral_constant<int, 3>{}, t + 3);
            if (t + 4 > nkt) break;
        }
        // no staging request may outlive the workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef A4_DMA_K
#undef A4_DMA_V

    // ---- epilogue: O^T out of the accumulator half (12 wait states after the last MFMA), normalise, store.
    // The lane's row is re-derived from a lane id the compiler cannot trace back, so that no output address is computed
    // ahead of the loop and carried through it (there is no register to carry it in).
    a4_nop<12>();
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r_e = lane_e & 31, h_e = lane_e >> 5;
    int w_e;
    asm volatile("s_mov_b32 %0, %1" : "=s"(w_e) : "s"(w));   // (the same for the wave index: the scalar copy is the one to keep)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float l_tot = pair32_sum(l_run[c]);
        const int qi_c = 64 * w_e + 32 * c + r_e;
        const bool live = (qi_c < q_rows);
        const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
        float o[MT][16];
#define A4_G4(c_, mt, b) o[mt][b] = a4_get<64 * c_ + 16 * mt + b>(); o[mt][b + 1] = a4_get<64 * c_ + 16 * mt + b + 1>(); \
                         o[mt][b + 2] = a4_get<64 * c_ + 16 * mt + b + 2>(); o[mt][b + 3] = a4_get<64 * c_ + 16 * mt + b + 3>();
#define A4_G16(c_, mt) A4_G4(c_, mt, 0) A4_G4(c_, mt, 4) A4_G4(c_, mt, 8) A4_G4(c_, mt, 12)
        if (c == 0) { A4_G16(0, 0) A4_G16(0, 1) A4_G16(0, 2) A4_G16(0, 3) }
        else { A4_G16(1, 0) A4_G16(1, 1) A4_G16(1, 2) A4_G16(1, 3) }
#undef A4_G16
#undef A4_G4
        if (live) {
            __bf16* optr = p.o + (int64_t)(o_row0 + qi_c) * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (__bf16)(o[mt][4 * g4 + e] * inv);
                    *reinterpret_cast<bf16x4*>(optr + 32 * mt + 8 * g4 + 4 * h_e) = ov;
                }
            if (h_e == 0) {
                float lse = (l_tot > 0.f) ? (m_run[c] * 0.6931471805599453f + __logf(l_tot)) : -INFINITY;
                p.lse[(int64_t)head * p.lse_stride + o_row0 + qi_c] = lse;
            }
        }
    }
    asm volatile("" :: "a"(hold[0]), "a"(hold[1]), "a"(hold[2]), "a"(hold[3]), "a"(hold[4]), "a"(hold[5]), "a"(hold[6]), "a"(hold[7]),
                       "a"(hold[8]), "a"(hold[9]), "a"(hold[10]), "a"(hold[11]), "a"(hold[12]), "a"(hold[13]), "a"(hold[14]), "a"(hold[15]));
}

int lv_launch_fwd_a4(const FwdParams& p0, int causal, hipStream_t stream) {
    FwdParams p = p0;
    p.head_per_xcd = lv_head_per_xcd(p) ? 1 : 0;
    dim3 grid(p.head_per_xcd ? (unsigned)(((p.hkv + 7) / 8) * 8 * p.n_blocks * p.group)
                             : (unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8 * p.group)), block(256);
    size_t lds = (size_t)(4 + 4) * 64 * 128 * 2;   // V ring + K ring
    if (causal) hipLaunchKernelGGL((lv_fwd_a4_kernel<true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_fwd_a4_kernel<false>), grid, block, lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
