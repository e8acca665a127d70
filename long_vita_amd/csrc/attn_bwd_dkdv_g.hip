// dK / dV kernel of the attention backward for gfx950, "one wave per SIMD, gap-scheduled" form (head dim 128;
// LVATTN_DKDV_KERNEL=g selects it).
//
// One workgroup = 4 waves = one 128-key block of one K/V head; wave w keeps the K and V fragments and the dK^T and dV^T
// accumulators of ITS 32 keys (key on the lane) in the 512 registers a lone wave of a SIMD has, while the workgroup
// sweeps "units" (a 32-row query slice of one query head of the group; slices last to first, heads inside), staged in
// an LDS ring by direct global -> LDS loads.  Same arithmetic and accumulation order as the first-generation kernel of
// attn_bwd.hip: S = Q K^T, P = exp2(c S - lse2), dV^T += dO^T P, dP = dO V^T, dS = P (dP - delta), dK^T += Q^T dS.
//
// What differs is the instruction stream.  A lone wave issues everything its SIMD runs, about one instruction per four
// cycles, and a v_mfma_f32_32x32x16_bf16 keeps the matrix pipe busy for 32: an iteration (one unit, 32 MFMAs, ONE
// workgroup barrier) is written as 32 "gaps", each holding its MFMA followed by a fixed share of the other work --
// exponentials, dS, bf16 packing, the LDS fragment reads of later gaps, the staging requests -- and a scheduling
// barrier after every gap keeps the compiler from packing the MFMAs together and the vector work behind them.
// Software pipeline of iteration u:
//   gaps  0- 7  dP(u)    = dO(u) V^T (onto -delta) | P(u) = exp2(c S(u) - lse2(u))   | read Q(u+1) rows (4-7)
//   gaps  8-15  S(u+1)   = Q(u+1) K^T              | pack P(u); dS(u) = P dP (12-19) | read dO^T(u) (from 12)
//   gaps 16-23  dV^T    += dO^T(u) P(u)            | dS(u), pack                     | read dO^T(u), Q^T(u), dO(u+1) rows
//   gaps 24-31  dK^T    += Q^T(u) dS(u)            | staging requests of unit u+5    | read Q^T(u), -lse2(u+1), -delta(u+1)
// The row constants (lse, natural log, and delta) come one loop trip -- six units -- per pair of requests into a small
// ring of their own; the wave that requested a block rewrites it in LDS as -lse * log2(e) and -delta before its first
// reader, so that P is one fused multiply-add and one exponential per score, and dS one multiply (the dP product starts
// from -delta instead of zero).
// The loop is unrolled over the six ring slots, so every LDS address is a lane register plus an immediate, and so is
// every decision about the constants (whose request it is, which iteration rescales, where a unit's constants lie).
//
// Register file, laid out by hand as in the one-wave forward kernel (attn_fwd_a4.hip): the accumulator half holds
//   a[0:63] dV^T (tile mt at 16 mt)   a[64:127] dK^T   a[128:159] K fragments   a[160:191] V fragments
//   a[192:223] Q rows of the next unit   a[224:255] dO rows of the current unit (both read from LDS straight into place)
// and is touched only by the asm statements here (MFMAs naming those registers, ds_read_b128 into them, the epilogue);
// the vector half (scores, P, dS, transposed fragments, row constants, addresses) is the compiler's.  Left to itself
// the allocator does not split such a kernel over the two halves: the plain-builtin form of this loop spilled 583
// registers.  The compiler neither counts nor pads what is inside an asm statement: LDS reads into the accumulator half
// are waited for by counted s_waitcnt inside the MFMA statement that needs them, and the wait states (an MFMA result ->
// a vector reader: 12; a vector write -> an MFMA operand: 2) hold by construction of the gap schedule (every such pair
// is at least three gaps apart).
#include "a4_asm.h"

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void g_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        g_static_for<I + 1, N>(f);
    }
}

// accumulator register numbers
__device__ constexpr int g_DV(int mt) { return 16 * mt; }
__device__ constexpr int g_DK(int mt) { return 64 + 16 * mt; }
__device__ constexpr int g_KF(int ks) { return 128 + 4 * ks; }
__device__ constexpr int g_VF(int ks) { return 160 + 4 * ks; }
__device__ constexpr int g_QA(int ks) { return 192 + 4 * ks; }
__device__ constexpr int g_OA(int ks) { return 224 + 4 * ks; }

// LDS operations of gap k that every wave issues (asm and compiler's alike) between the Q-row reads and their use: the
// Q rows themselves (two per gap in gaps 4-7), the transposed dO fragments (two 8-byte reads per gap from gap 12 on)
__device__ constexpr int g_lds_ops(int k) { return (k >= 4 && k < 8) ? 2 : (k >= 12 && k < 16) ? 2 : 0; }
// LDS operations that may still be in flight when the MFMA of gap 8 + j takes Q-row fragment j, read in gap 4 + j / 2:
// all those of the gaps after that one (the order inside a gap is the compiler's, so none of its own gap is counted)
__device__ constexpr int g_qa_wait(int j) {
    int n = 0;
    for (int k = 4 + j / 2 + 1; k < 8 + j; ++k) n += g_lds_ops(k);
    return n > 15 ? 15 : n;
}

// x (vector registers) = or += A fragment x B fragment, both in the accumulator half; WAIT >= 0: first let all but WAIT
// LDS reads land
template <int A, int B, bool FIRST, int WAIT>
__device__ __forceinline__ void g_mfma_x(f32x16& x) {
    if constexpr (WAIT >= 0) asm volatile("s_waitcnt lgkmcnt(%c0)" :: "i"(WAIT) : "memory");
    if constexpr (FIRST)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(x) : "i"(A), "i"(A + 3), "i"(B), "i"(B + 3));
    else
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(x) : "i"(A), "i"(A + 3), "i"(B), "i"(B + 3));
}

// Direct global -> LDS loads (lane i -> LDS byte lds_dst + 16 i; lds_dst in M0).  As asm so that the compiler
// does not count them; completion is waited for by hand.
// M0 is not restored: nothing else in this kernel uses it.
#ifndef G_PROBE
#define G_PROBE 0
#endif
__device__ __forceinline__ void g_dma16(a4_u32x4 rs, int voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}
template <int A, int N>
__device__ __forceinline__ void g_get(float* o) {
    o[0] = a4_get<A>();
    if constexpr (N > 1) g_get<A + 1, N - 1>(o + 1);
}

}  // namespace

// Timing probes (wrong results): a build with -DG_PROBE=<bits> leaves out one class of the loop's work
#define G_OPAQUE(x) asm volatile("" : "=v"(x))

template <bool CAUSAL, bool ACCUM_F32>
__global__ __launch_bounds__(256) void lv_bwd_dkdv_g_kernel(BwdParams p) {
    constexpr int PROBE = G_PROBE;
    constexpr int D = 128;
    constexpr int BQ = 32;             // query rows per unit
    constexpr int KS = D / 16;         // 8
    constexpr int MT = D / 32;         // 4
    constexpr int ROWB = D * 2;
    constexpr int CPR = D / 8;
    constexpr int QTILE = BQ * ROWB;   // 8 KiB: a unit's Q (or dO) rows
    constexpr int SLOTB = 2 * QTILE;   // [Q rows][dO rows]
    constexpr int CBLKB = 2048;        // row constants of the six units of one loop trip: [6][32] x -lse2 at +0, [6][32] x -delta at +1024 (f32)
    constexpr int NCB = 3;             // ring of such blocks: the trip's, the next trip's, the one on its way
    constexpr int CST0 = 6 * SLOTB;    // ... behind the unit slots
    constexpr int NBUF = 6;            // ring slots: units u-? .. u+5 around iteration u
    constexpr int AHEAD = NBUF - 1;    // staging distance in units
    constexpr int HALF = 3;            // slots per address register set (ds offsets are 16-bit)
    constexpr int RPP = 64 / CPR;      // rows per 1-KiB piece: 4
    constexpr int PPW = 4;             // pieces per wave per unit (16 pieces: 8 of Q, 8 of dO)

    // The accumulator half belongs to the asm statements below: sixteen 16-register values of the accumulator class are
    // declared here and kept alive to the end, so that the register allocator counts all 256 as occupied and parks
    // nothing of its own there (which value it maps to which register does not matter: it never touches them).
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

    // Workgroup i runs on XCD i % 8 (own L2 each): key blocks are dealt to the XCDs by KV head, as in the role-split kernel
    const int unit_id = blockIdx.x;
    if (unit_id >= p.n_blocks * p.hkv) return;
    const int blk = unit_id / p.hkv;
    const int kvh = unit_id % p.hkv;

    const AttnDesc dsc = p.desc[blk];
    const int k_row0 = dsc.a_row0, k_rows = dsc.a_rows;
    const int q_row0 = dsc.b_row0, q_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;
    const int nslice = (q_len + BQ - 1) / BQ;
    const int niter = nslice * p.group;

    // ---- K / V fragments -> registers (B operands; this lane's key = 32 w + r)
    const int kbase_w = 32 * w;
    const int kj = kbase_w + r;
    {
        const int grow = k_row0 + (kj < k_rows ? kj : k_rows - 1);
        const __bf16* kp = p.k + (int64_t)grow * p.k_rs + (int64_t)kvh * p.k_hs + 8 * h;
        const __bf16* vp = p.v + (int64_t)grow * p.v_rs + (int64_t)kvh * p.v_hs + 8 * h;
        a4_u32x4 f[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const a4_u32x4*>(kp + 16 * ks);
        g_static_for<0, KS>([&](auto I_) __attribute__((always_inline)) { a4_put4<g_KF(decltype(I_)::value)>(f[decltype(I_)::value]); });
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = *reinterpret_cast<const a4_u32x4*>(vp + 16 * ks);
        g_static_for<0, KS>([&](auto I_) __attribute__((always_inline)) { a4_put4<g_VF(decltype(I_)::value)>(f[decltype(I_)::value]); });
        __builtin_amdgcn_sched_barrier(0);
    }
    // dV^T = dK^T = 0
    a4_zero16<0>(); a4_zero16<16>(); a4_zero16<32>(); a4_zero16<48>(); a4_zero16<64>(); a4_zero16<80>(); a4_zero16<96>(); a4_zero16<112>();

    // ---- staging of a unit: waves 0, 1 bring the Q rows, waves 2, 3 the dO rows; piece j of wave w is rows
    // RPP (PPW (w & 1) + j) .. + RPP - 1.  The LDS image is XOR-swizzled inside each row; the LDS side of the transfer is
    // lane-linear, so the swizzle goes on the source.
    const bool w_is_do = (w >= 2);
    const int64_t x_rs = w_is_do ? p.do_rs : p.q_rs;
    const int64_t x_hs = w_is_do ? p.do_hs : p.q_hs;
    const __bf16* const x_base = (w_is_do ? p.dout : p.q) + (int64_t)q_row0 * x_rs;
    const int prow0 = RPP * PPW * (w & 1) + lane / CPR;
    const int rows_last = q_len - (nslice - 1) * BQ;   // only the last slice (the first one requested) can be partial
    int voff_full[PPW], voff_part[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        const int ch16 = lds_off128(row, lane % CPR) - row * ROWB;
        voff_full[j] = row * (int)(x_rs * 2) + ch16;
        voff_part[j] = min(row, rows_last - 1) * (int)(x_rs * 2) + ch16;   // rows past the end repeat the last valid one
    }
    // The request stream: (slice, head) of the next unit to request, walked with as few scalar instructions as it takes (a
    // lone wave pays four cycles for each).  Only the first slice requested (the last of the sweep) can be partial: its
    // offsets and record count are the stream's initial state and are replaced when the stream leaves that slice.  Past
    // the last unit the stream keeps walking, with zero records: nothing is fetched, the slot is one nobody reads.
    int voff_cur[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) voff_cur[j] = voff_part[j];
    const int rec_full = (BQ - 1) * (int)(x_rs * 2) + ROWB;
    int rec_cur = (rows_last - 1) * (int)(x_rs * 2) + ROWB;
    int ld_hg = 0, ld_n = 0;                    // head of the next request inside its group; requests so far
    const char* ld_x = reinterpret_cast<const char*>(x_base + (int64_t)kvh * p.group * x_hs + (int64_t)(nslice - 1) * BQ * x_rs);
    const int64_t x_step_head = x_hs * 2;
    const int64_t x_step_slice = -((int64_t)(p.group - 1) * x_hs + (int64_t)BQ * x_rs) * 2;
    // Row constants (lse and delta of a unit's 32 rows) come one loop trip -- six units -- at a time: a direct-to-LDS
    // request costs the wave that issues it about sixty cycles whatever it carries, and the other three waves then
    // wait for it at the barrier (one 256-byte request per unit was 4.5 % of the kernel).  Two 1-KiB requests, one per
    // array, bring a trip's constants: lane = 8 (unit in the trip) + (group of four rows), 16 bytes each, lanes 48..63
    // repeat the last unit into padding.  The whole [hq, lse_stride] arrays are the buffers: a group of rows past the
    // end of a segment reads whatever lies there (those rows are masked, and their dP sanitised, below), one past the
    // end of the array reads zero.  Block b is requested by wave b mod 4 two trips ahead, in the trip's first
    // iteration, so whose turn it is and where a unit's constants lie are compile-time facts of the unrolled loop.
    a4_u32x4 rs_lse, rs_delta;
    {
        const uint32_t nrec = (uint32_t)((int64_t)p.hq * p.lse_stride * 4);   // < 2^32: the launcher checks
        const uint64_t lb = (uint64_t)(uintptr_t)p.lse, db = (uint64_t)(uintptr_t)p.delta;
        rs_lse[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lb);
        rs_lse[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(lb >> 32)) & 0xffffu;
        rs_lse[2] = nrec; rs_lse[3] = 0x00020000u;
        rs_delta[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)db);
        rs_delta[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(db >> 32)) & 0xffffu;
        rs_delta[2] = nrec; rs_delta[3] = 0x00020000u;
    }
    const int64_t c_row0 = (int64_t)kvh * p.group * p.lse_stride + q_row0 + 4 * (lane & 7);   // + head * lse_stride + 32 * slice
    // A wave's blocks are w, w + 4, w + 8, ...: the lane keeps (unit number, slices back from the last one, head in the
    // group) of its unit of the wave's NEXT block and steps them by 24 units after each request, so that no request
    // pays for a division.
    int cn_unit = 6 * w + ((lane >> 3) < 6 ? (lane >> 3) : 5);
    int cn_sl = cn_unit / p.group, cn_hg = cn_unit - cn_sl * p.group;
    const int cn_sl24 = 24 / p.group, cn_hg24 = 24 - cn_sl24 * p.group;
    const int cn_sl_last = (niter - 1) / p.group, cn_hg_last = (niter - 1) - cn_sl_last * p.group;
    auto dma_const_block = [&](int cslot) __attribute__((always_inline)) {
        const bool past = cn_unit >= niter;        // past the last unit: the last one again
        const int sl_back = past ? cn_sl_last : cn_sl, hg = past ? cn_hg_last : cn_hg;
        const int voff = (int)((c_row0 + (int64_t)hg * p.lse_stride + (int64_t)(nslice - 1 - sl_back) * BQ) * 4);
        const uint32_t dst = lds_base + CST0 + cslot * CBLKB;
        g_dma16(rs_lse, voff, dst);
        g_dma16(rs_delta, voff, dst + CBLKB / 2);
        cn_unit += 24;
        cn_sl += cn_sl24;
        cn_hg += cn_hg24;
        if (cn_hg >= p.group) { cn_hg -= p.group; ++cn_sl; }
    };
    // the request of one unit, in three parts so that they can sit in different gaps: the descriptor is made a gap
    // before the pieces (a scalar register written by a vector instruction needs five wait states before a memory
    // instruction reads it, and nothing pads an asm statement)
    a4_u32x4 ld_rs;
    auto dma_prepare = [&]() __attribute__((always_inline)) {
        // word by word and said to be wave-uniform (left to itself the compiler does the 64-bit pointer arithmetic of
        // the request stream in vector registers, which a buffer instruction cannot take)
        const uint64_t xb = (uint64_t)(uintptr_t)ld_x;
        ld_rs[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)xb);
        ld_rs[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(xb >> 32)) & 0xffffu;
        ld_rs[2] = (uint32_t)__builtin_amdgcn_readfirstlane(ld_n < niter ? rec_cur : 0);
        ld_rs[3] = 0x00020000u;
        asm volatile("" : "+s"(ld_rs));      // made here, not next to its first use
    };
    auto dma_pieces = [&](int slot) __attribute__((always_inline)) {
        const uint32_t dst = lds_base + slot * SLOTB + (w_is_do ? QTILE : 0) + PPW * (w & 1) * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j) g_dma16(ld_rs, voff_cur[j], dst + j * 1024);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        ++ld_n;
        if (++ld_hg == p.group) {       // on to the slice before
            ld_hg = 0;
            ld_x += x_step_slice;
            rec_cur = rec_full;
#pragma unroll
            for (int j = 0; j < PPW; ++j) voff_cur[j] = voff_full[j];
        } else {
            ld_x += x_step_head;
        }
    };

    // ---- LDS read addresses: one register set for slots 0..2 and one for slots 3..5 (ds offsets are 16-bit); the slot and
    // the operand (Q at +0, dO at +QTILE) are immediates.  Row reads: the XOR swizzle acts on address
    // bits 4..7 only and every slot starts on a multiple of 256 bytes, so fragment ks (chunk 2 ks + h) is the address of
    // fragment 0 with ks XORed into bits 5..7.
    uint32_t a_addr[2];
    a_addr[0] = lds_base + (uint32_t)lds_off128(r, h);
    a_addr[1] = a_addr[0] + HALF * SLOTB;
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* t_ptr[2][MT][2];       // transposed reads (rows in accumulator order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            t_ptr[0][mt][jh] = lds0 + lds_off128(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);
            t_ptr[1][mt][jh] = t_ptr[0][mt][jh] + HALF * SLOTB;
        }
    // row constants of rows 8 g + 4 h .. + 3 of unit i of a trip: block base + 128 i + 32 g (-lse2), + 1024 more (-delta)
    lds_char* const c_lane = lds0 + CST0 + 16 * h;
    lds_char* c_blk = c_lane;                 // this trip's block
    lds_char* c_blk_next = c_lane + CBLKB;    // the next trip's
    int c_slot = 0;                           // ring slot of this trip's block; the trip number mod 4 says whose turn a request is
    int trip_w = 0;

    const float c = p.scale_log2e;

    // pipeline state carried from one iteration to the next (the dO rows of the current unit are in a[224:255])
    f32x16 sv;            // S of the current unit until gap 7, of the next one from gap 8
    f32x16 dp;            // -delta of the current unit's rows at the start of an iteration: the dP product accumulates onto it
    f32x4 l4[4];          // the current unit's -lse * log2(e), rows 8 g + 4 h .. + 3

    // The staged row constants are lse (natural log) and delta.  What the loop wants is -lse * log2(e) (so that P is one
    // fused multiply-add and one exponential per score) and -delta (the initial accumulator of the dP product, so that dS
    // is one multiply): the wave that requested a block rewrites it in place, 16 + 16 bytes per lane, in the third
    // iteration of the trip before the block's (barriers lie between that and the first reader).
    lds_char* const resc_ptr = lds0 + CST0 + 16 * lane;
    auto rescale_load = [&](int cslot, f32x4& a, f32x4& b) __attribute__((always_inline)) {
        a = *(LDS_PTR(f32x4, resc_ptr + cslot * CBLKB));
        b = *(LDS_PTR(f32x4, resc_ptr + cslot * CBLKB + CBLKB / 2));
    };
    auto rescale_store = [&](int cslot, f32x4& a, f32x4& b) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] *= -1.4426950408889634f; b[e] = -b[e]; }
        *(LDS_PTR(f32x4, resc_ptr + cslot * CBLKB)) = a;
        *(LDS_PTR(f32x4, resc_ptr + cslot * CBLKB + CBLKB / 2)) = b;
    };

    int c_sl = nslice - 1, c_hg = 0;   // (slice, head) being computed
    // A unit needs the mask when its first row i0 is below the wave's last key's diagonal (kbase_w + 31 > i0 + shift), when
    // its rows run past the segment (i0 + BQ > q_len), or always when the wave's keys run past the key block.
    const int mask_below = (kbase_w + 32 > k_rows) ? 0x7fffffff : kbase_w + 31 - shift;
    const int mask_from = q_len - BQ + 1;

    // One iteration.  SLOT = u mod 6 fixes every ring slot at compile time.
    auto iteration = [&](auto SLOT_) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(SLOT_)::value;
        constexpr int NEXT = (SLOT + 1) % NBUF;
        constexpr int DMAS = (SLOT + AHEAD) % NBUF;
        constexpr int CS = SLOT / HALF, CO = (SLOT % HALF) * SLOTB;      // address set and byte offset of the current slot
        constexpr int NS = NEXT / HALF, NO = (NEXT % HALF) * SLOTB;      // ... of the next unit's slot

        const int i0 = c_sl * BQ;   // first query row of the unit, relative to q_row0
        if (++c_hg == p.group) { c_hg = 0; --c_sl; }
        const bool need_mask = (i0 < mask_below) || (i0 >= mask_from);

        // Everything older than the last two iterations' requests has landed (this wave's share of unit u+2); the
        // barrier makes that true of every wave's share, and tells this wave that all waves are done with the slot of
        // unit u-1, which this iteration's requests overwrite.  No LDS read is left in flight either (the dO rows and
        // the lse of this unit were read at the end of the last iteration).
        // (the wave that requested a block of constants in the trip's first iteration has those two requests in flight
        // in the next two: counting them in keeps the wait from reaching into the last two iterations' pieces)
        const bool creq_mine = (((trip_w + 2) & 3) == w);       // this trip's request is this wave's
        if constexpr (PROBE & 128) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if constexpr (PROBE & 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PPW) : "memory");
        else if ((SLOT == 1 || SLOT == 2) && creq_mine) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PPW + 2) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PPW) : "memory");
        __builtin_amdgcn_sched_barrier(0);

        bf16x8 doT[2 * MT], qT[2 * MT];   // transposed fragments (s2, mt) of this unit, each alive for about four gaps
        float pv[16];
        // constants: the trip's first iteration requests the block two trips on, its third rescales the next trip's
        const bool resc_mine = (((trip_w + 1) & 3) == w);
        const int c_slot1 = (c_slot == NCB - 1) ? 0 : c_slot + 1, c_slot2 = (c_slot1 == NCB - 1) ? 0 : c_slot1 + 1;
        f32x4 resc_a, resc_b;
        lds_char* const c_rd = (SLOT == NBUF - 1 ? c_blk_next : c_blk) + ((SLOT + 1) % NBUF) * (BQ * 4);   // unit u+1's constants
        bf16x8 pf[2], dsf[2];

        auto tr_read = [&](bf16x8& f, int set, int off, int mt) __attribute__((always_inline)) {
            if constexpr (PROBE & 16) { G_OPAQUE(f); return; }
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[set][mt][0] + off));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[set][mt][1] + off));
            f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        };

        g_static_for<0, 32>([&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;
            // ---- the MFMA of this gap
            if constexpr (k < 8) {
                g_mfma_x<g_OA(k), g_VF(k), false, -1>(dp);       // onto -delta
            } else if constexpr (k < 16) {
                g_mfma_x<g_QA(k - 8), g_KF(k - 8), k == 8, g_qa_wait(k - 8)>(sv);
            } else if constexpr (k < 24) {
                constexpr int j = k - 16, s2 = j / MT, mt = j % MT;
                a4_mfma_o<g_DV(mt), -1>(doT[j], pf[s2]);
            } else {
                constexpr int j = k - 24, s2 = j / MT, mt = j % MT;
                a4_mfma_o<g_DK(mt), -1>(qT[j], dsf[s2]);
            }
            __builtin_amdgcn_sched_barrier(0);      // the MFMA first, its gap's fillers behind it

            // ---- LDS reads, each about four gaps (128 cycles) ahead of its use: the LDS counter holds 15 requests, and a
            // wave that has more in flight stalls at the next one
            if constexpr (k >= 4 && k < 8 && !(PROBE & 8)) {        // Q rows of unit u+1 -> a[192:223], two fragments per gap, for gaps 8-15
                a4_read_k<g_QA(2 * (k - 4)), NO>(a_addr[NS] ^ (uint32_t)((2 * (k - 4)) << 5));
                a4_read_k<g_QA(2 * (k - 4) + 1), NO>(a_addr[NS] ^ (uint32_t)((2 * (k - 4) + 1) << 5));
            }
            if constexpr (SLOT == 2 && k == 0 && !(PROBE & 64)) { if (resc_mine) rescale_load(c_slot1, resc_a, resc_b); }
            if constexpr (SLOT == 2 && k == 3 && !(PROBE & 64)) { if (resc_mine) rescale_store(c_slot1, resc_a, resc_b); }
            if constexpr (k >= 12 && k < 20) {      // dO^T of unit u, fragment k - 12, needed in gap 16 + (k - 12)
                constexpr int j = k - 12, s2 = j / MT, mt = j % MT;
                tr_read(doT[j], CS, CO + QTILE + 16 * s2 * ROWB, mt);
            }
            if constexpr (k >= 20 && k < 28) {      // Q^T of unit u, fragment k - 20, needed in gap 24 + (k - 20)
                constexpr int j = k - 20, s2 = j / MT, mt = j % MT;
                tr_read(qT[j], CS, CO + 16 * s2 * ROWB, mt);
            }
            if constexpr (k >= 16 && k < 24 && !(PROBE & 8)) {      // dO rows of unit u+1 -> a[224:255] (read by the dP product of gaps 0-7, long done)
                a4_read_k<g_OA(k - 16), NO + QTILE>(a_addr[NS] ^ (uint32_t)((k - 16) << 5));
            }
            if constexpr (k >= 24 && k < 28 && !(PROBE & 64)) {      // -lse2 of unit u+1 (this unit's was last read in gap 7)
                l4[k - 24] = *(LDS_PTR(f32x4, c_rd + 32 * (k - 24)));
            }
            if constexpr (k >= 28 && !(PROBE & 64)) {                // -delta of unit u+1: where its dP product starts (dP of this unit was last read in gap 19)
                const f32x4 t4 = *(LDS_PTR(f32x4, c_rd + CBLKB / 2 + 32 * (k - 28)));
                dp[4 * (k - 28)] = t4[0]; dp[4 * (k - 28) + 1] = t4[1]; dp[4 * (k - 28) + 2] = t4[2]; dp[4 * (k - 28) + 3] = t4[3];
            }

            // ---- P(u): two scores per gap in gaps 0-7, the (rare) mask in gap 8, packed in gaps 8-11
            if constexpr (k < 8) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * k + e;
                    if constexpr (PROBE & 32) G_OPAQUE(pv[i]);
                    else if constexpr (PROBE & 1) pv[i] = __builtin_fmaf(sv[i], c, l4[i >> 2][i & 3]);
                    else pv[i] = fast_exp2(__builtin_fmaf(sv[i], c, l4[i >> 2][i & 3]));
                }
            }
            if constexpr (k == 8) {
                if (need_mask) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        const bool dead = (kj > ii + shift) | (ii >= q_len) | (kj >= k_rows);
                        pv[i] = dead ? 0.f : pv[i];
                        asm volatile("" : "+v"(pv[i]));   // keeps this a branch: as selects it would run for every unit
                    }
                    // a row past the end of the segment read whatever lay behind the segment's constants: its dP product
                    // started from anything, NaN included (the product's last MFMA is a gap old: wait its result out)
                    if (i0 >= mask_from) {
                        a4_nop<16>(); a4_nop<16>();
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            dp[i] = (ii >= q_len) ? 0.f : dp[i];
                            asm volatile("" : "+v"(dp[i]));
                        }
                    }
                }
            }
            if constexpr (k >= 8 && k < 12 && (PROBE & 32)) { if constexpr (k & 1) G_OPAQUE(pf[(k - 8) >> 1]); }
            if constexpr (k >= 8 && k < 12 && !(PROBE & 32)) {
                constexpr int i = 4 * (k - 8);
                pf[i >> 3][(i & 7)] = (__bf16)pv[i]; pf[i >> 3][(i & 7) + 1] = (__bf16)pv[i + 1];
                pf[i >> 3][(i & 7) + 2] = (__bf16)pv[i + 2]; pf[i >> 3][(i & 7) + 3] = (__bf16)pv[i + 3];
                if constexpr (k == 9 || k == 11) asm volatile("" : "+v"(pf[(k - 8) >> 1]));   // formed here, not next to its use
            }
            // ---- dS(u) = P (dP - delta): two values per gap in gaps 12-19 (the product started from -delta and is complete
            // after gap 7), packed in 20-23
            if constexpr (k >= 12 && k < 20 && !(PROBE & 32)) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * (k - 12) + e;
                    pv[i] = pv[i] * dp[i];
                }
            }
            if constexpr (k >= 20 && k < 24 && (PROBE & 32)) { if constexpr (k & 1) G_OPAQUE(dsf[(k - 20) >> 1]); }
            if constexpr (k >= 20 && k < 24 && !(PROBE & 32)) {
                constexpr int i = 4 * (k - 20);
                dsf[i >> 3][(i & 7)] = (__bf16)pv[i]; dsf[i >> 3][(i & 7) + 1] = (__bf16)pv[i + 1];
                dsf[i >> 3][(i & 7) + 2] = (__bf16)pv[i + 2]; dsf[i >> 3][(i & 7) + 3] = (__bf16)pv[i + 3];
                if constexpr (k == 21 || k == 23) asm volatile("" : "+v"(dsf[(k - 20) >> 1]));
            }
            // ---- staging requests of unit u+5 (its slot held unit u-1, which every wave finished before this
            // iteration's barrier)
            if constexpr (k == 23 && !(PROBE & 4)) dma_prepare();
            if constexpr (k == 24 && !(PROBE & 4)) dma_pieces(DMAS);
            if constexpr (k == 28 && !(PROBE & 4)) {
                if constexpr (SLOT == 0 && !(PROBE & 256)) { if (creq_mine) dma_const_block(c_slot2); }   // (any gap does: 2, 12 and 28 measured the same)
                dma_advance();
            }
            if constexpr (SLOT == NBUF - 1 && k == 31) {      // on to the next trip
                trip_w = (trip_w + 1) & 3;
                c_slot = c_slot1;
                c_blk = c_blk_next;
                c_blk_next = c_lane + c_slot2 * CBLKB;
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    if (niter > 0) {
        // prologue: units 0 .. 4 requested and landed; S(0), then the dO rows and the lse of unit 0
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) { dma_prepare(); a4_nop<4>(); dma_pieces(i); dma_advance(); }
        if (w < 2) dma_const_block(w);         // the constants of trips 0 and 1 (trip 0 requests those of trip 2)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (w == 0) {      // (trip 1's block is rescaled by the loop, in trip 0's third iteration, like every later one)
            f32x4 ta, tb;
            rescale_load(0, ta, tb);
            rescale_store(0, ta, tb);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        g_static_for<0, KS>([&](auto I_) __attribute__((always_inline)) {
            constexpr int ks = decltype(I_)::value;
            a4_read_k<g_QA(ks), 0>(a_addr[0] ^ (uint32_t)(ks << 5));
            a4_read_k<g_OA(ks), QTILE>(a_addr[0] ^ (uint32_t)(ks << 5));
        });
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            l4[g] = *(LDS_PTR(f32x4, c_lane + 32 * g));
            const f32x4 t4 = *(LDS_PTR(f32x4, c_lane + CBLKB / 2 + 32 * g));
            dp[4 * g] = t4[0]; dp[4 * g + 1] = t4[1]; dp[4 * g + 2] = t4[2]; dp[4 * g + 3] = t4[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        g_static_for<0, KS>([&](auto I_) __attribute__((always_inline)) {
            constexpr int ks = decltype(I_)::value;
            g_mfma_x<g_QA(ks), g_KF(ks), ks == 0, -1>(sv);
        });
        a4_nop<12>();
        __builtin_amdgcn_sched_barrier(0);
        for (int it = 0;; it += NBUF) {
            iteration(std::integral_constant<int, 0>{});
            if (it + 1 >= niter) break;
            iteration(std::integral_constant<int, 1>{});
            if (it + 2 >= niter) break;
            iteration(std::integral_constant<int, 2>{});
            if (it + 3 >= niter) break;
            iteration(std::integral_constant<int, 3>{});
            if (it + 4 >= niter) break;
            iteration(std::integral_constant<int, 4>{});
            if (it + 5 >= niter) break;
            iteration(std::integral_constant<int, 5>{});
            if (it + 6 >= niter) break;
        }
    }
    // no staging request may outlive the workgroup: its LDS could already belong to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: dK^T / dV^T out of the accumulator half (12 wait states after the last MFMA; key on the lane) ->
    // rows of dk / dv
    a4_nop<12>();
    float dvv[MT][16], dkk[MT][16];
    g_get<0, 64>(&dvv[0][0]);
    g_get<64, 64>(&dkk[0][0]);
    if (kj < k_rows) {
        const int64_t grow = k_row0 + dsc.out_delta + kj;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * mt + 8 * g4 + 4 * h;
                if constexpr (ACCUM_F32) {
                    float* dkp = reinterpret_cast<float*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                    float* dvp = reinterpret_cast<float*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                    f32x4 a = p.f32_store ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(dkp);
                    f32x4 b = p.f32_store ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(dvp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] += dkk[mt][4 * g4 + e] * p.scale; b[e] += dvv[mt][4 * g4 + e]; }
                    *reinterpret_cast<f32x4*>(dkp) = a;
                    *reinterpret_cast<f32x4*>(dvp) = b;
                } else {
                    __bf16* dkp = reinterpret_cast<__bf16*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                    __bf16* dvp = reinterpret_cast<__bf16*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                    bf16x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(dkk[mt][4 * g4 + e] * p.scale); b[e] = (__bf16)dvv[mt][4 * g4 + e]; }
                    *reinterpret_cast<bf16x4*>(dkp) = a;
                    *reinterpret_cast<bf16x4*>(dvp) = b;
                }
            }
    }
    asm volatile("" :: "a"(hold[0]), "a"(hold[1]), "a"(hold[2]), "a"(hold[3]), "a"(hold[4]), "a"(hold[5]), "a"(hold[6]), "a"(hold[7]),
                       "a"(hold[8]), "a"(hold[9]), "a"(hold[10]), "a"(hold[11]), "a"(hold[12]), "a"(hold[13]), "a"(hold[14]), "a"(hold[15]));
}

int lv_launch_bwd_dkdv_g(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    if (p.n_blocks <= 0) return 0;
    dim3 grid((unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8)), block(256);
    if ((int64_t)p.hq * p.lse_stride * 4 > 0xffffffffLL) return -3;   // the row constants are addressed with 32-bit byte offsets
    const size_t lds = (size_t)6 * (2 * 32 * 128 * 2) + 3 * 2048;   // unit slots + the constants' ring
    if (causal) {
        if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_g_kernel<true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dkdv_g_kernel<true, false>), grid, block, lds, stream, p);
    } else {
        if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_g_kernel<false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dkdv_g_kernel<false, false>), grid, block, lds, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
