// Flash-style attention forward for gfx950, "one wave per SIMD" form (head dim 128; LVATTN_FWD_KERNEL=a4).
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
// The accumulator registers are named as clobbers once so that the kernel descriptor allocates them; the build must
// show no compiler v_accvgpr_* and no spill (the vector half has to stay within 256 registers).
//
// The loop is software-pipelined: iteration t holds O^T += V_{t-1}^T P_{t-1}^T, S_{t+1}^T = K_{t+1} Q^T and the softmax of
// S_t, the vector work spread over the 64 MFMA gaps of the iteration; one workgroup barrier per 64-key tile.
// K / V tiles live in a 3-slot LDS ring (slot i % 3 = [K tile i][V tile i]); iteration t requests K tile t+3 and V tile
// t+1 by direct global -> LDS loads (each wave a quarter of each), a counted s_waitcnt vmcnt before the barrier leaves
// the newest iteration's requests in flight.
#include "internal.h"
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((ext_vector_type(4))) uint32_t a4_u32x4;

#define A4_CLOBBER_ALL                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18",  \
        "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35",  \
        "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52",  \
        "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69",  \
        "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86",  \
        "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102",     \
        "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", \
        "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", \
        "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", \
        "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", \
        "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", \
        "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", \
        "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", \
        "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", \
        "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", \
        "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", \
        "a253", "a254", "a255"

__device__ __forceinline__ void a4_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// accumulator register numbers
__device__ constexpr int a4_O(int c, int mt) { return 64 * c + 16 * mt; }
__device__ constexpr int a4_Q(int c, int ks) { return 128 + 32 * c + 4 * ks; }
__device__ constexpr int a4_K(int st, int ks) { return 192 + 32 * st + 4 * ks; }

// a[A .. A+3] = one 16-byte fragment held in vector registers
template <int A>
__device__ __forceinline__ void a4_put4(a4_u32x4 x) {
    asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
                 :: "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "i"(A), "i"(A + 1), "i"(A + 2), "i"(A + 3));
}
template <int A>
__device__ __forceinline__ void a4_zero16() {
    asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "i"(A));
    if constexpr ((A & 15) != 15) a4_zero16<A + 1>();
}
template <int A>
__device__ __forceinline__ float a4_get() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "i"(A));
    return x;
}
// a[A] *= alpha for 16 consecutive registers (a rare path: the running maximum grew by more than the threshold)
template <int A, int N>
__device__ __forceinline__ void a4_scale(float alpha) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a[%c2]\n\ts_nop 0\n\tv_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_accvgpr_write_b32 a[%c2], %0"
                 : "=&v"(t) : "v"(alpha), "i"(A));
    if constexpr (N > 1) a4_scale<A + 1, N - 1>(alpha);
}
// K fragment (st, ks) <- LDS
template <int A, int OFF>
__device__ __forceinline__ void a4_read_k(uint32_t lds_addr) {
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" :: "v"(lds_addr), "i"(A), "i"(A + 3), "i"(OFF) : "memory");
}
// S^T tile += K fragment x Q fragment (both in the accumulator half); FIRST: the product starts from zero
template <int KA, int QA, bool FIRST>
__device__ __forceinline__ void a4_mfma_s(f32x16& s) {
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "i"(KA), "i"(KA + 3), "i"(QA), "i"(QA + 3));
}
// O^T tile += V fragment x P fragment (vector registers)
template <int OA>
__device__ __forceinline__ void a4_mfma_o(bf16x8 vf, bf16x8 pf) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(vf), "v"(pf), "i"(OA), "i"(OA + 15));
}
// two fp32 -> one dword of two bf16 (round to nearest even), the first in the low half
__device__ __forceinline__ uint32_t a4_pack(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    bf16x2_ v;
    v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(uint32_t, v);
}
template <int N>
__device__ __forceinline__ void a4_nop() { asm volatile("s_nop %c0" :: "i"(N - 1)); }

template <int I, int N, typename F>
__device__ __forceinline__ void a4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        a4_static_for<I + 1, N>(f);
    }
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
    constexpr int NS = 3;
    constexpr int SLOTB = 2 * TILEB;
    constexpr int RPP = 64 / CPR;     // rows per 1-KiB piece: 4
    constexpr int PPW = 4;            // pieces per wave per tile and operand
    constexpr float RESCALE_LOG2 = 8.0f;
    constexpr float OPTIMISTIC_MAX = 4096.0f;

    // the accumulator half belongs to the asm statements below
    asm volatile("" ::: A4_CLOBBER_ALL);

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
    const int q_row0 = dsc.a_row0, q_rows = dsc.a_rows;
    const int k_row0 = dsc.b_row0, k_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;

    // ---- Q fragments of the two column blocks into a[128:191]: lane holds Q[row][16ks + 8h .. +8]
    int qi[2], limit[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        qi[c] = 64 * w + 32 * c + r;
        limit[c] = qi[c] + shift;
        if (limit[c] > k_len - 1 || qi[c] >= q_rows) limit[c] = k_len - 1;
    }
    {
        a4_u32x4 qv[2][KS];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int q_row = q_row0 + (qi[c] < q_rows ? qi[c] : q_rows - 1);
            const __bf16* qptr = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * h;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qv[c][ks] = *reinterpret_cast<const a4_u32x4*>(qptr + 16 * ks);
        }
#define A4_PUTQ(c, ks) a4_put4<a4_Q(c, ks)>(qv[c][ks]);
        A4_PUTQ(0, 0) A4_PUTQ(0, 1) A4_PUTQ(0, 2) A4_PUTQ(0, 3) A4_PUTQ(0, 4) A4_PUTQ(0, 5) A4_PUTQ(0, 6) A4_PUTQ(0, 7)
        A4_PUTQ(1, 0) A4_PUTQ(1, 1) A4_PUTQ(1, 2) A4_PUTQ(1, 3) A4_PUTQ(1, 4) A4_PUTQ(1, 5) A4_PUTQ(1, 6) A4_PUTQ(1, 7)
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
    const int64_t k_tile_bytes = (int64_t)BN * p.k_rs * 2, v_tile_bytes = (int64_t)BN * p.v_rs * 2;
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);
    const int prow0 = RPP * PPW * w + lane / CPR;
    int row_ch[PPW];   // piece j: row (bits 8..) and the swizzled 16-byte position inside the row (low bits)
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        row_ch[j] = (row << 8) | (lds_off128(row, lane % CPR) - row * ROWB);
    }
    // piece j of K tile t (isv = 0) or V tile t (isv = 1) into its ring slot
    auto issue_piece = [&](int t, int isv, int j) __attribute__((always_inline)) {
        int rows = rows_avail - t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        const int64_t rs_ = isv ? p.v_rs : p.k_rs;
        const uint32_t rec = rows ? (uint32_t)((rows - 1) * rs_ * 2 + ROWB) : 0u;
        const char* base = isv ? vbase + t * v_tile_bytes : kbase + t * k_tile_bytes;
        // (the descriptor is rebuilt from scalar halves so that it is provably wave-uniform)
        const uint64_t b64 = (uint64_t)(uintptr_t)base;
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b64 >> 32));
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b64);
        const char* ubase = (const char*)(uintptr_t)(((uint64_t)b_hi << 32) | (uint64_t)b_lo);
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ubase), 0, (uint32_t)__builtin_amdgcn_readfirstlane((int)rec), 0x00020000);
        const int last = rows - 1;
        const uint32_t dst = lds_base + (uint32_t)((t % NS) * SLOTB + (isv ? TILEB : 0) + (PPW * w + j) * 1024);
        const int row = min(row_ch[j] >> 8, last < 0 ? 0 : last);   // rows past the end repeat the last valid one
        a4_dma16(rsrc, row * (int)(rs_ * 2) + (row_ch[j] & 255), dst);
    };

    // ---- LDS read addresses in slot 0
    uint32_t k_addr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_addr[ks] = lds_base + (uint32_t)lds_off128(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* v_ptr[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            v_ptr[mt][jh] = lds0 + TILEB + lds_off128(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);

    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    const float cs = p.scale_log2e;
    f32x16 sA[2][2], sB[2][2];        // S^T of two consecutive tiles: [column block][32-key half]
    a4_u32x4 pA[2][2][2], pB[2][2][2];  // P^T (bf16 pairs) of two consecutive tiles: [column block][half][16-key quarter]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < 4; ++i) { pA[c][st][s2][i] = 0u; pB[c][st][s2][i] = 0u; }

    // exact softmax of tile t held in sc (mask, row maximum, deferred rescale) -> pf; updates m_run / l_run / O
    auto softmax_exact = [&](f32x16 (&sc)[2][2], a4_u32x4 (&pf)[2][2][2], int t, bool need_mask) __attribute__((always_inline)) {
        const int kb0 = t * BN;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (need_mask) {
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int j = kb0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h;
                        sc[c][st][i] = (j > limit[c]) ? -INFINITY : sc[c][st][i];
                    }
            }
            float mx = sc[c][0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, sc[c][0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, sc[c][1][i]);
            mx = pair32_max(mx);
            const float m_new = fmaxf(m_run[c], mx);
            const bool grow = (m_new - m_run[c]) * cs > RESCALE_LOG2;
            if (__any(grow)) {
                const float alpha = fast_exp2((m_run[c] - m_new) * cs);
                m_run[c] = m_new;
                l_run[c] *= alpha;
                if (c == 0) a4_scale<0, 64>(alpha); else a4_scale<64, 64>(alpha);
            }
            const float mc = m_run[c] * cs;
            float rs = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float pa = fast_exp2(__builtin_fmaf(sc[c][st][i], cs, -mc));
                    const float pb = fast_exp2(__builtin_fmaf(sc[c][st][i + 1], cs, -mc));
                    rs += pa + pb;
                    pf[c][st][i >> 3][(i & 7) >> 1] = a4_pack(pa, pb);
                }
            l_run[c] += rs;
        }
    };

    // One iteration (one barrier): 64 MFMAs, each followed by its share of the other work ("gap" k = 0 .. 63):
    //   gaps  0-31  O^T += V_{t-1}^T P_{t-1}^T   group g = k / 8 = (st, s2), then (mt, c)
    //   gaps 32-63  S_{t+1}^T = K_{t+1} Q^T       segment (st, kh), then (ks, c)
    //   every gap   softmax of S_t -> P_t against the running maximum (optimistic pass): even gaps the two exponentials
    //               of a pair of scores, odd gaps the row sums and the cast; a tile that needs masking or whose sums
    //               come out too large is redone exactly after the gaps (S_t is still intact)
    //   gaps  0-23  V fragments of the next group;  gaps 16-31  K_{t+1} fragments -> a[192:255];  8 staging requests
    auto iteration = [&](int t, f32x16 (&scur)[2][2], f32x16 (&snext)[2][2], a4_u32x4 (&pprev)[2][2][2], a4_u32x4 (&pcur)[2][2][2])
                         __attribute__((always_inline)) {
        // everything older than the previous iteration's requests has landed (K tile t+1, V tile t-1)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        const int vb = ((t + NS - 1) % NS) * SLOTB;     // slot of V tile t-1 (t = 0: P is zero, whatever is read)
        const uint32_t kb = (uint32_t)(((t + 1) % NS) * SLOTB);   // slot of K tile t+1
        const bool active = (t < my_nkt);
        const int kb0 = t * BN;
        const bool need_mask = (kb0 + BN - 1 > 64 * w + shift) || (kb0 + BN > k_len);
        const bool fast = active && !need_mask && !__any(m_run[0] < -1e29f || m_run[1] < -1e29f);
        const float mc0 = m_run[0] * cs, mc1 = m_run[1] * cs;
        float rs0 = 0.f, rs1 = 0.f;
        float pvA = 0.f, pvB = 0.f;
        bf16x8 vfr[2][MT];
        auto load_v = [&](int set, int g) __attribute__((always_inline)) {
            const int voff = vb + (32 * (g >> 1) + 16 * (g & 1)) * ROWB;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][0] + voff));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][1] + voff));
                bf16x8& vf = vfr[set][mt];
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
            }
        };
        load_v(0, 0);
        a4_static_for<0, 64>([&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;
            // ---- the MFMA of this gap
            if constexpr (k < 32) {
                constexpr int g = k >> 3, st = g >> 1, s2 = g & 1, mt = (k & 7) >> 1, c = k & 1;
                if constexpr ((k & 7) == 0 && g < 3) load_v((g + 1) & 1, g + 1);      // the next group's fragments
                a4_mfma_o<a4_O(c, mt)>(vfr[g & 1][mt], __builtin_bit_cast(bf16x8, pprev[c][st][s2]));
            } else {
                constexpr int sg = (k - 32) >> 3, st = sg >> 1, kh = sg & 1, ks = 4 * kh + ((k & 7) >> 1), c = k & 1;
                if constexpr (k == 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K_{t+1} fragments are in
                a4_mfma_s<a4_K(st, ks), a4_Q(c, ks), ks == 0>(snext[c][st]);
            }
            __builtin_amdgcn_sched_barrier(0);      // the MFMA first, its gap's fillers behind it
            // ---- K_{t+1} fragments, one per gap
            if constexpr (k >= 16 && k < 32) {
                constexpr int f = k - 16, st = f >> 3, ks = f & 7;
                a4_read_k<a4_K(st, ks), st * 32 * ROWB>(k_addr[ks] + kb);
            }
            // ---- staging requests: K tile t+3 (pieces 0-3), V tile t+1 (pieces 0-3)
            if constexpr ((k & 7) == 5) {
                constexpr int j = k >> 3;
                if constexpr (j < 4) issue_piece(t + 3, 0, j); else issue_piece(t + 1, 1, j - 4);
            }
            // ---- softmax share: pair e of the 32 (two scores of S_t per lane)
            {
                constexpr int e = k >> 1, c = e >> 4, st = (e >> 3) & 1, i0 = 2 * (e & 7);
                if constexpr ((k & 1) == 0) {
                    pvA = fast_exp2(__builtin_fmaf(scur[c][st][i0], cs, c ? -mc1 : -mc0));
                    pvB = fast_exp2(__builtin_fmaf(scur[c][st][i0 + 1], cs, c ? -mc1 : -mc0));
                } else {
                    // (the empty asm statements keep the sums and the cast in THIS gap: left alone, the compiler gathers
                    // them at the end of the iteration and keeps all 64 exponentials alive until then)
                    if constexpr (c) { rs1 += pvA; rs1 += pvB; asm volatile("" : "+v"(rs1)); }
                    else { rs0 += pvA; rs0 += pvB; asm volatile("" : "+v"(rs0)); }
                    uint32_t pw = a4_pack(pvA, pvB);
                    asm volatile("" : "+v"(pw));
                    pcur[c][st][i0 >> 3][(i0 & 7) >> 1] = pw;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        bool exact = active && !fast;
        if (fast) {
            if (__any(!(rs0 <= OPTIMISTIC_MAX) || !(rs1 <= OPTIMISTIC_MAX))) exact = true;
            else { l_run[0] += rs0; l_run[1] += rs1; }
        }
        if (exact) softmax_exact(scur, pcur, t, need_mask);
        if (!active) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int i = 0; i < 4; ++i) pcur[c][st][s2][i] = 0u;
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nkt > 0) {
        // prologue: K tiles 0, 1, 2 and V tile 0, then S_0
#pragma unroll
        for (int j = 0; j < PPW; ++j) { issue_piece(0, 0, j); issue_piece(1, 0, j); issue_piece(2, 0, j); issue_piece(0, 1, j); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define A4_RK(st, ks) a4_read_k<a4_K(st, ks), st * 32 * ROWB>(k_addr[ks]);
        A4_RK(0, 0) A4_RK(0, 1) A4_RK(0, 2) A4_RK(0, 3) A4_RK(0, 4) A4_RK(0, 5) A4_RK(0, 6) A4_RK(0, 7)
        A4_RK(1, 0) A4_RK(1, 1) A4_RK(1, 2) A4_RK(1, 3) A4_RK(1, 4) A4_RK(1, 5) A4_RK(1, 6) A4_RK(1, 7)
#undef A4_RK
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define A4_S(c, st, ks, first) a4_mfma_s<a4_K(st, ks), a4_Q(c, ks), first>(sA[c][st]);
#define A4_S8(c, st) A4_S(c, st, 0, true) A4_S(c, st, 1, false) A4_S(c, st, 2, false) A4_S(c, st, 3, false) \
                     A4_S(c, st, 4, false) A4_S(c, st, 5, false) A4_S(c, st, 6, false) A4_S(c, st, 7, false)
        A4_S8(0, 0) A4_S8(1, 0) A4_S8(0, 1) A4_S8(1, 1)
#undef A4_S8
#undef A4_S
        a4_nop<12>();
        // iteration t: PV of tile t-1, S of tile t+1, softmax of tile t; one more iteration for the last PV
        for (int t = 0; t <= nkt; t += 2) {
            iteration(t, sA, sB, pB, pA);
            if (t + 1 <= nkt) iteration(t + 1, sB, sA, pA, pB);
        }
        // no staging request may outlive the workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // ---- epilogue: O^T out of the accumulator half (12 wait states after the last MFMA), normalise, store
    a4_nop<12>();
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float l_tot = pair32_sum(l_run[c]);
        const bool live = (qi[c] < q_rows);
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
            __bf16* optr = p.o + (int64_t)(q_row0 + qi[c]) * p.o_rs + (int64_t)head * p.o_hs;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ov[e] = (__bf16)(o[mt][4 * g4 + e] * inv);
                    *reinterpret_cast<bf16x4*>(optr + 32 * mt + 8 * g4 + 4 * h) = ov;
                }
            if (h == 0) {
                float lse = (l_tot > 0.f) ? (m_run[c] * p.scale + __logf(l_tot)) : -INFINITY;
                p.lse[(int64_t)head * p.lse_stride + q_row0 + qi[c]] = lse;
            }
        }
    }
}

int lv_launch_fwd_a4(const FwdParams& p0, int causal, hipStream_t stream) {
    FwdParams p = p0;
    p.head_per_xcd = lv_head_per_xcd(p) ? 1 : 0;
    dim3 grid(p.head_per_xcd ? (unsigned)(((p.hkv + 7) / 8) * 8 * p.n_blocks * p.group)
                             : (unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8 * p.group)), block(256);
    size_t lds = (size_t)3 * 2 * 64 * 128 * 2;
    if (causal) hipLaunchKernelGGL((lv_fwd_a4_kernel<true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_fwd_a4_kernel<false>), grid, block, lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
