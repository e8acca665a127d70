// Flash-style attention backward for gfx950 (MI355X), no atomics.  This file holds the delta kernel,
// the launcher, and the first generation of the two gradient kernels; the launcher uses the faster
// second generation by default (attn_bwd_dkdv_rs.hip: role-split dK/dV, attn_bwd_dq_pp.hip: ping-pong
// dQ) and these only when LVATTN_DKDV_KERNEL=single / LVATTN_DQ_KERNEL=lockstep ask for them.
//
// lv_bwd_delta_kernel : delta[h][row] = sum_d dO * O   (fp32)
//
// lv_bwd_dkdv_kernel  : one workgroup = 4 waves = one 128-key block of one K/V head; wave w
//   keeps the K, V fragments and the dK^T, dV^T accumulators of its 32 keys in registers (key on the lane) while the workgroup
//   sweeps 32-row query slices (last to first) x the query heads of the group, staged in LDS:
//     S  = Q K^T        (A = Q rows from LDS,  B = K fragments in registers)
//     P  = exp2(S*c - lse*log2e)
//     dV^T += dO^T P    (A = dO read column-wise with ds_read_b64_tr_b16, B = P registers)
//     dP = dO V^T       (A = dO rows,          B = V fragments in registers)
//     dS = P * (dP - delta) * scale      (P taken back from its bf16 form)
//     dK^T += Q^T dS    (A = Q read column-wise, B = dS registers)
//   P and dS go from the accumulator of one MFMA straight into the B operand of the next.
//
// lv_bwd_dq_kernel    : one workgroup = 8 waves = one 256-row query block of one query head
//   (the forward kernel's shape; query on the lane), sweeping 64-key K/V tiles staged in LDS:
//     S^T  = K Q^T      (A = K rows,  B = Q fragments in registers)
//     dP^T = V dO^T     (A = V rows,  B = dO fragments in registers)
//     dS^T = exp2(S^T*c - lse*log2e) * (dP^T - delta)
//     dQ^T += K^T dS^T  (A = K read column-wise, B = dS^T registers)
//   and dQ = scale * dQ^T is written once.
// S and P are recomputed in both kernels (7 matrix products instead of 5); in exchange no
// gradient is summed across workgroups, so results are bitwise reproducible.
#include "internal.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_BWD_RING
#define LV_BWD_RING 4   // LDS slots of the dK/dV kernel's Q / dO staging ring
#endif
#ifndef LV_BWD_KT
#define LV_BWD_KT 1   // key sub-tiles per wave of the dK/dV kernel; LV_BWD_BLOCK_N = 128 * LV_BWD_KT
#endif
static_assert(128 * LV_BWD_KT == LV_BWD_BLOCK_N_INTERNAL, "keep in step with LV_BWD_BLOCK_N in lvattn.h");

// Direct global -> LDS loads (no register destination): lane i's 16 (or 4) bytes land at LDS byte
// address lds_dst + 16 i (or 4 i); lds_dst is wave-uniform and travels in M0.  Written as asm so that
// the compiler does not count them: it would otherwise wait for ALL of them before the next
// transposed LDS read.  Their completion is waited for by hand (counted s_waitcnt vmcnt + barrier).
__device__ __forceinline__ void dma16_buffer(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4_global(const float* g, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}

template <int D>
__device__ __forceinline__ int lds_offd(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

// ---------------------------------------------------------------------------
__global__ void lv_bwd_delta_kernel(const __bf16* __restrict__ o, const __bf16* __restrict__ dout, float* __restrict__ delta,
                                    int rows, int hq, int d, int64_t o_rs, int64_t o_hs, int64_t do_rs, int64_t do_hs,
                                    int64_t lse_stride) {
    const int cpr = d / 8;  // threads per (row, head): 16 or 8
    int64_t total = (int64_t)rows * hq * cpr;
    int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // grid covers `total` rounded up to the block size; all lanes stay active for the shuffles
    int64_t cidx = idx < total ? idx : total - 1;
    int ch = (int)(cidx % cpr);
    int64_t rh = cidx / cpr;
    int head = (int)(rh % hq);
    int64_t row = rh / hq;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(o + row * o_rs + (int64_t)head * o_hs + ch * 8);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(dout + row * do_rs + (int64_t)head * do_hs + ch * 8);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)a[i] * (float)b[i];
    for (int off = cpr / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (ch == 0 && idx < total) delta[(int64_t)head * lse_stride + row] = s;
}

int lv_launch_bwd_delta(const void* o, const void* dout, float* delta, int rows, int hq, int d, int64_t o_rs, int64_t o_hs,
                        int64_t do_rs, int64_t do_hs, int64_t lse_stride, hipStream_t stream) {
    int64_t total = (int64_t)rows * hq * (d / 8);
    int64_t grid = (total + 255) / 256;
    hipLaunchKernelGGL(lv_bwd_delta_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const __bf16*)o, (const __bf16*)dout, delta, rows,
                       hq, d, o_rs, o_hs, do_rs, do_hs, lse_stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------
// dK / dV
// ---------------------------------------------------------------------------
template <int D, bool CAUSAL, bool ACCUM_F32>
__global__ __launch_bounds__(256) void lv_bwd_dkdv_kernel(BwdParams p) {
    constexpr int NW = 4;              // one wave per SIMD: the whole 512-register file per wave
    constexpr int KT = LV_BWD_KT;      // 32-key sub-tiles per wave (32 keys per wave, 128 per workgroup)
    static_assert(KT == 1, "the software-pipelined loop below handles one key sub-tile per wave");
    constexpr int BQ = 32;             // query rows per slice
    constexpr int KS = D / 16;         // k-steps over d
    constexpr int MT = D / 32;         // 32-wide d sub-tiles
    constexpr int ROWB = D * 2;        // bytes per Q/dO row in LDS
    constexpr int CPR = D / 8;         // 16-byte chunks per row
    constexpr int QTILE = BQ * ROWB;   // bytes of a Q (or dO) slice
    // LDS ring of NBUF slots.  While slice `it` is computed on, slice it+1 is read for S, and the
    // slices up to it+NBUF-1 are in flight from memory.  One slot:
    //   [Q slice][dO slice][4 x (32 lse, 32 delta) f32: one copy per wave]
    constexpr int NBUF = LV_BWD_RING;
    constexpr int AHEAD = NBUF - 1;            // staging distance in slices
    constexpr int SLOTB = 2 * QTILE + NW * 256;
    // Staging is done by direct global->LDS loads; each wave-instruction writes 1 KiB of LDS
    // (lane i -> bytes 16i..16i+15), i.e. RPP whole rows of the image.
    constexpr int RPP = 64 / CPR;                  // rows per 1-KiB piece: 4 (d=128) or 8 (d=64)
    constexpr int PPW = (2 * QTILE / 1024) / NW;   // pieces per wave per slice: 4 or 2
    constexpr int LPI = PPW + 1;                   // loads per wave per slice (pieces + the lse/delta row)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds0 = (lds_char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int bid = blockIdx.x;
    const int blk = bid % p.n_blocks;
    const int kvh = bid / p.n_blocks;

    const AttnDesc dsc = p.desc[blk];
    const int k_row0 = dsc.a_row0, k_rows = dsc.a_rows;
    const int q_row0 = dsc.b_row0, q_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;
    const int nslice = (q_len + BQ - 1) / BQ;
    const int niter = nslice * p.group;

    // ---- K / V fragments -> registers (B operands; this lane's key = 32w + r)
    const int kbase_w = 32 * w;
    const int kj = kbase_w + r;
    bf16x8 kf[KS], vf[KS];
    {
        int grow = k_row0 + (kj < k_rows ? kj : k_rows - 1);
        const __bf16* kp = p.k + (int64_t)grow * p.k_rs + (int64_t)kvh * p.k_hs + 8 * h;
        const __bf16* vp = p.v + (int64_t)grow * p.v_rs + (int64_t)kvh * p.v_hs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
        }
    }
    // retire these loads now: inside the loop the only memory counter traffic is the staging
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(kf[ks])); asm volatile("" : "+v"(vf[ks])); }

    // ---- staging of a (slice, head).  Waves 0,1 bring the Q rows, waves 2,3 the dO rows; piece j
    // of wave w is rows RPP*(PPW*(w&1) + j) .. +RPP-1.  The LDS image is XOR-swizzled inside each
    // row, and as the LDS side of the transfer is lane-linear the swizzle goes on the source:
    // lane i of a piece holds position (i % CPR) of row (i / CPR), which must receive the global
    // chunk (i % CPR) ^ f(row).
    const bool w_is_do = (w >= NW / 2);
    const int64_t x_rs = w_is_do ? p.do_rs : p.q_rs;
    const int64_t x_hs = w_is_do ? p.do_hs : p.q_hs;
    const __bf16* const x_base = (w_is_do ? p.dout : p.q) + (int64_t)q_row0 * x_rs;
    const int prow0 = RPP * PPW * (w & 1) + lane / CPR;  // row of this lane in piece 0
    int pch16[PPW], voff_full[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        // position -> chunk is the same involution as chunk -> position
        pch16[j] = (lds_offd<D>(row, lane % CPR) - row * ROWB);
        voff_full[j] = row * (int)(x_rs * 2) + pch16[j];
    }
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds0);
    // Iteration order: query slices from the LAST one down to the first, and for each slice the
    // query heads of the group.  Key blocks of one sequence all end at the same query row, so
    // workgroups that start together read the same Q / dO rows at the same time (L2 reuse).
    // The request stream walks (slice, head) pairs in the sweep's order; its source addresses move
    // by a constant step from one request to the next (next head, or first head of the slice
    // before), so they are kept as running pointers: no multiplications per request.
    int ld_sl = nslice - 1, ld_hg = 0;  // (slice, head) of the next staging request
    uint32_t ld_slot = lds_base;        // LDS byte address of the ring slot it goes to
    int ld_n = 0;                       // requests issued so far
    const char* ld_x = reinterpret_cast<const char*>(x_base + (int64_t)kvh * p.group * x_hs + (int64_t)ld_sl * BQ * x_rs);
    const int64_t x_step_head = x_hs * 2;
    const int64_t x_step_slice = -((int64_t)(p.group - 1) * x_hs + (int64_t)BQ * x_rs) * 2;
    // row constants: lanes 0..31 fetch lse of rows 0..31, lanes 32..63 delta of rows 0..31
    const float* ld_c = ((lane < 32) ? p.lse : p.delta) + q_row0 + (int64_t)kvh * p.group * p.lse_stride + (int64_t)ld_sl * BQ + r;
    const int64_t c_step_head = p.lse_stride;
    const int64_t c_step_slice = -((int64_t)(p.group - 1) * p.lse_stride + BQ);
    // only the last slice of the sequence (the first one requested) can be partial: rows past the
    // end repeat the last valid row
    const int rows_last = q_len - (nslice - 1) * BQ;
    int voff_part[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) voff_part[j] = min(prow0 + RPP * j, rows_last - 1) * (int)(x_rs * 2) + pch16[j];
    const int c_part = min(r, rows_last - 1) - r;
    auto issue_dma = [&]() __attribute__((always_inline)) {
        const bool partial = (ld_sl == nslice - 1);  // (a full last slice has voff_part == voff_full)
        const int rows_valid = partial ? rows_last : BQ;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(ld_x), 0, (uint32_t)((rows_valid - 1) * x_rs * 2 + ROWB), 0x00020000);
        const uint32_t dst = ld_slot + (w_is_do ? QTILE : 0) + PPW * (w & 1) * 1024;
#pragma unroll
        for (int j = 0; j < PPW; ++j) dma16_buffer(rs, partial ? voff_part[j] : voff_full[j], dst + j * 1024);
        dma4_global(ld_c + (partial ? c_part : 0), ld_slot + 2 * QTILE + w * 256);  // every wave keeps its own copy
        // advance to the next request; past the last slice the stream stays where it is, so the
        // loop's unconditional requests re-fetch that slice into a slot nobody reads any more
        const bool adv = (++ld_n < niter);
        const bool wrap = (ld_hg + 1 == p.group);
        const int64_t xs = wrap ? x_step_slice : x_step_head, cs = wrap ? c_step_slice : c_step_head;
        ld_x += adv ? xs : 0;
        ld_c += adv ? cs : 0;
        ld_hg = adv ? (wrap ? 0 : ld_hg + 1) : ld_hg;
        ld_sl = (adv && wrap) ? ld_sl - 1 : ld_sl;
        ld_slot = (ld_slot + SLOTB == lds_base + NBUF * SLOTB) ? lds_base : ld_slot + SLOTB;
    };

    // ---- LDS read addresses in slot 0 (the slot offset is added per iteration, dO is an immediate)
    lds_char* a_ptr[KS];  // row reads of Q: row r, chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_ptr[ks] = lds0 + lds_offd<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* t_ptr[MT][2];  // transposed reads of Q (rows = queries in accumulator order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            t_ptr[mt][jh] = lds0 + lds_offd<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);
    // this wave's copy of the row constants: lse of rows 8g+4h .. +3 and delta of the same rows
    lds_char* const ld_ptr = lds0 + 2 * QTILE + w * 256 + 16 * h;
    // and the same copy seen one float per lane (lanes 0..31 lse, 32..63 delta), for the rescaling pass
    lds_char* const ld_lane_ptr = lds0 + 2 * QTILE + w * 256 + 4 * lane;
    const float ld_lane_mul = (lane < 32) ? 1.4426950408889634f : 1.f;

    f32x16 dk[MT], dv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[mt][i] = 0.f; dv[mt][i] = 0.f; }
    const float c = p.scale_log2e;

    // The staged lse is in natural-log units; the wave rewrites ITS copy as lse * log2(e) (one float
    // per lane) before the first use, instead of 16 multiplications per lane in the product loop.
    auto rescale_lse = [&](int sb) __attribute__((always_inline)) {
        float v = *(LDS_PTR(float, ld_lane_ptr + sb));
        *(LDS_PTR(float, ld_lane_ptr + sb)) = v * ld_lane_mul;
    };
    // LDS -> register fragment loads for the slot at byte offset sb
    auto load_rows = [&](bf16x8 (&f)[KS], int sb) __attribute__((always_inline)) {  // A operand: 32 rows x d
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = *(LDS_PTR(bf16x8, a_ptr[ks] + sb));
    };
    auto load_cols = [&](bf16x8 (&f)[2 * MT], int sb) __attribute__((always_inline)) {  // A operand: d x 32 rows
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][0] + sb + 16 * s2 * ROWB));
                bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][1] + sb + 16 * s2 * ROWB));
                bf16x8& af = f[s2 * MT + mt];
                af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
            }
    };
    auto load_consts = [&](f32x4 (&v)[4], int sb, int which) __attribute__((always_inline)) {  // which: 0 lse2, 1 delta
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = *(LDS_PTR(f32x4, ld_ptr + sb + 32 * g + 128 * which));
    };
    // Wait until only the newest AHEAD-2 staging requests of this wave are in flight, then the
    // workgroup barrier.  (A plain __syncthreads() would wait for ALL of them: staging loads are
    // pending LDS writes.)
    auto wait_and_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((AHEAD - 2) * LPI) : "memory");
    };

    // prologue: slices 0 .. AHEAD-1 requested, 0 and 1 landed
    if (niter > 0) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) issue_dma();
        wait_and_barrier();
    }

    // ---- the sweep.  One wave per SIMD: nothing but the wave's own instruction stream can keep
    // the matrix pipe busy, so every product of 8 MFMAs runs on fragments that were read from LDS
    // during the product before it, beside the vector work of its neighbours:
    //   A  dP(it)   = dO V^T     | P = exp2(c S(it) - lse2)    | read dO^T(it), delta(it)
    //   B  dV      += dO^T P     | dS = P (dP - delta)         | read Q(it+1) rows
    //   C  S(it+1)  = Q K^T      | next staging request        | read Q^T(it)
    //   D  dK      += Q^T dS     |                             | read dO(it+1) rows, lse2(it+1)
    // (dS lacks the factor `scale`, which the epilogue applies to dK.)  Every wave computes every
    // slice, also the few next to the diagonal that the mask empties for its keys: a second,
    // shorter code path for them would make the compiler copy the 128 accumulator registers
    // between the paths on every iteration.
    int c_sl = nslice - 1, c_hg = 0;  // (slice, head) being computed
    int c_sb = 0;                     // byte offset of its ring slot
    bf16x8 oa[KS];   // dO rows of the current slice
    f32x4 l4[4];     // its lse2
    f32x16 sv;       // its S
    if (niter > 0) {
        rescale_lse(0);
        bf16x8 qa[KS];
        load_rows(qa, 0);
        load_rows(oa, QTILE);
        load_consts(l4, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], sv, 0, 0, 0);
    }
    for (int it = 0; it < niter; ++it) {
        const int i0 = c_sl * BQ;  // first query row of the slice, relative to q_row0
        if (++c_hg == p.group) { c_hg = 0; --c_sl; }
        const int sb = c_sb;        // byte offset of this slice's slot
        c_sb = (c_sb + SLOTB == NBUF * SLOTB) ? 0 : c_sb + SLOTB;
        const int nsb = c_sb;       // and of the next slice's (after the last slice: a slot nobody reads any more)
        const bool need_mask = (kbase_w + 31 > i0 + shift) || (i0 + BQ > q_len) || (kbase_w + 32 > k_rows);

        // ---- A
        bf16x8 ta[2 * MT];
        f32x4 d4[4];
        rescale_lse(nsb);
        load_cols(ta, sb + QTILE);
        load_consts(d4, sb, 1);
        f32x16 dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa[ks], vf[ks], dp, 0, 0, 0);
        float pv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) pv[i] = fast_exp2(__builtin_fmaf(sv[i], c, -l4[i >> 2][i & 3]));
        if (need_mask) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                bool dead = (kj > ii + shift) | (ii >= q_len) | (kj >= k_rows);
                pv[i] = dead ? 0.f : pv[i];
                asm volatile("" : "+v"(pv[i]));  // keeps this a branch: as selects it would run for every slice
            }
        }
        bf16x8 pf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) pf[i >> 3][i & 7] = (__bf16)pv[i];
        asm volatile("" : "+v"(pf[0]), "+v"(pf[1]));  // P is formed here, not next to its first use
        __builtin_amdgcn_sched_barrier(0);

        // ---- B
        bf16x8 qa[KS];
        load_rows(qa, nsb);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[s2 * MT + mt], pf[s2], dv[mt], 0, 0, 0);
        bf16x8 dsf[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) dsf[i >> 3][i & 7] = (__bf16)(pv[i] * (dp[i] - d4[i >> 2][i & 3]));
        asm volatile("" : "+v"(dsf[0]), "+v"(dsf[1]));  // dS is formed here, beside the dV products
        __builtin_amdgcn_sched_barrier(0);

        // ---- C
        load_cols(ta, sb);
#pragma unroll
        for (int i = 0; i < 16; ++i) sv[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks], kf[ks], sv, 0, 0, 0);
        // the slot this request goes to held slice it-1, which every wave finished with before the last barrier
        issue_dma();
        __builtin_amdgcn_sched_barrier(0);

        // ---- D
        load_rows(oa, nsb + QTILE);
        load_consts(l4, nsb, 0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ta[s2 * MT + mt], dsf[s2], dk[mt], 0, 0, 0);
        // slice it+2 must have landed before the next iteration reads it; the requests issued
        // after it (it+3 .. it+AHEAD) stay in flight across the barrier
        wait_and_barrier();
    }
    // no staging request may outlive the workgroup: its LDS could already belong to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: dK^T / dV^T accumulators (key on the lane) -> rows of dk / dv
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
                    for (int e = 0; e < 4; ++e) { a[e] += dk[mt][4 * g4 + e] * p.scale; b[e] += dv[mt][4 * g4 + e]; }
                    *reinterpret_cast<f32x4*>(dkp) = a;
                    *reinterpret_cast<f32x4*>(dvp) = b;
                } else {
                    __bf16* dkp = reinterpret_cast<__bf16*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                    __bf16* dvp = reinterpret_cast<__bf16*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                    bf16x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)(dk[mt][4 * g4 + e] * p.scale); b[e] = (__bf16)dv[mt][4 * g4 + e]; }
                    *reinterpret_cast<bf16x4*>(dkp) = a;
                    *reinterpret_cast<bf16x4*>(dvp) = b;
                }
            }
    }
}

// ---------------------------------------------------------------------------
// dQ
// ---------------------------------------------------------------------------
template <int D, bool CAUSAL, bool ACCUM_F32>
__global__ __launch_bounds__(512, 2) void lv_bwd_dq_kernel(BwdParams p) {
    constexpr int NW = 8;
    constexpr int NT = NW * 64;
    constexpr int BN = 64;            // keys per tile
    constexpr int KS = D / 16;
    constexpr int MT = D / 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = BN * ROWB;
    constexpr int CPR = D / 8;
    constexpr int NLD = (BN * CPR) / NT;  // 16-byte loads per thread per tile and matrix

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K buf0][K buf1][V buf0][V buf1]
    lds_char* const ldsK = (lds_char*)smem;
    constexpr int VOFF = 2 * TILEB;  // V tiles follow the K tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int bid = blockIdx.x;
    const int hg = bid % p.group;
    const int rest = bid / p.group;
    const int blk = rest % p.n_qblocks;
    const int kvh = rest / p.n_qblocks;
    const int head = kvh * p.group + hg;

    const AttnDesc dsc = p.qdesc[blk];
    const int q_row0 = dsc.a_row0, q_rows = dsc.a_rows;
    const int k_row0 = dsc.b_row0, k_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;

    // ---- Q and dO fragments (B operands): lane holds row r, columns 16ks + 8h .. +8
    const int qi = 32 * w + r;
    const int q_row = q_row0 + (qi < q_rows ? qi : q_rows - 1);
    bf16x8 qf[KS], dof[KS];
    {
        const __bf16* qp = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * h;
        const __bf16* op = p.dout + (int64_t)q_row * p.do_rs + (int64_t)head * p.do_hs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
            dof[ks] = *reinterpret_cast<const bf16x8*>(op + 16 * ks);
        }
    }
    float lse2 = p.lse[(int64_t)head * p.lse_stride + q_row];
    lse2 = (lse2 == -INFINITY) ? INFINITY : lse2 * 1.4426950408889634f;
    const float delta = p.delta[(int64_t)head * p.lse_stride + q_row];

    int limit = qi + shift;
    if (limit > k_len - 1 || qi >= q_rows) limit = k_len - 1;
    int wave_kmax = 32 * w + 32 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    const int my_nkt = (32 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging through buffer loads (rows past the key range read as zero)
    const char* kbase = reinterpret_cast<const char*>(p.k + (int64_t)kvh * p.k_hs + (int64_t)k_row0 * p.k_rs);
    const char* vbase = reinterpret_cast<const char*>(p.v + (int64_t)kvh * p.v_hs + (int64_t)k_row0 * p.v_rs);
    const int64_t k_tile_bytes = (int64_t)BN * p.k_rs * 2, v_tile_bytes = (int64_t)BN * p.v_rs * 2;
    int st_lds[NLD], st_koff[NLD], st_voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int cidx = tid + NT * i;
        int row = cidx / CPR, ch = cidx % CPR;
        st_lds[i] = lds_offd<D>(row, ch);
        st_koff[i] = (int)(row * p.k_rs * 2) + ch * 16;
        st_voff[i] = (int)(row * p.v_rs * 2) + ch * 16;
    }
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);
    u32x4 kreg[NLD], vreg[NLD];
    auto issue_loads = [&](int t) {
        int rows = rows_avail - t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        uint32_t krec = rows ? (uint32_t)((rows - 1) * p.k_rs * 2 + ROWB) : 0u;
        uint32_t vrec = rows ? (uint32_t)((rows - 1) * p.v_rs * 2 + ROWB) : 0u;
        auto krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase + t * k_tile_bytes), 0, krec, 0x00020000);
        auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase + t * v_tile_bytes), 0, vrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, st_koff[i], 0, 0));
            vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, st_voff[i], 0, 0));
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *(LDS_PTR(u32x4, ldsK + buf * TILEB + st_lds[i])) = kreg[i];
            *(LDS_PTR(u32x4, ldsK + VOFF + buf * TILEB + st_lds[i])) = vreg[i];
        }
    };

    // ---- LDS read addresses (K buffer 0; V and the other buffer are immediates)
    lds_char* k_ptr[KS];  // row reads: row r of the sub-tile, chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = ldsK + lds_offd<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* kt_ptr[MT][2];  // transposed reads of K: rows = keys in accumulator order, columns = d
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            kt_ptr[mt][jh] = ldsK + lds_offd<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);

    f32x16 dq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[mt][i] = 0.f;
    const float c = p.scale_log2e;

    if (nkt > 0) {
        issue_loads(0);
        write_lds(0);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(qf[ks])); asm volatile("" : "+v"(dof[ks])); }
    __syncthreads();

    auto tile = [&](int t, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const bool more = (t + 1 < nkt);
        if (more) issue_loads(t + 1);
        if (t < my_nkt) {
            const int kb0 = t * BN;
            const bool need_mask = (kb0 + BN - 1 > 32 * w + shift) || (kb0 + BN > k_len);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const int sub = cur * TILEB + st * 32 * ROWB;
                // S^T = K Q^T and dP^T = V dO^T for 32 keys
                f32x16 s, dp;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 kfr = *(LDS_PTR(bf16x8, k_ptr[ks] + sub));
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[ks], s, 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 vfr = *(LDS_PTR(bf16x8, k_ptr[ks] + VOFF + sub));
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, dof[ks], dp, 0, 0, 0);
                }
                // dS^T = P^T * (dP^T - delta)
                bf16x8 dsf[2];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float pv = fast_exp2(__builtin_fmaf(s[i], c, -lse2));
                    if (need_mask) {
                        int j = kb0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h;
                        pv = (j > limit) ? 0.f : pv;
                    }
                    dsf[i >> 3][i & 7] = (__bf16)(pv * (dp[i] - delta));
                }
                // dQ^T += K^T dS^T
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kt_ptr[mt][0] + sub + 16 * s2 * ROWB));
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kt_ptr[mt][1] + sub + 16 * s2 * ROWB));
                        bf16x8 af;
                        af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                        af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
                        dq[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf[s2], dq[mt], 0, 0, 0);
                    }
            }
        }
        if (more) write_lds(cur ^ 1);
        __syncthreads();
    };
    for (int t = 0; t < nkt; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nkt) tile(t + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: dQ^T accumulators (query on the lane) -> rows of dq
    if (qi < q_rows) {
        const int64_t grow = q_row0 + dsc.out_delta + qi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * mt + 8 * g4 + 4 * h;
                if constexpr (ACCUM_F32) {
                    float* dqp = reinterpret_cast<float*>(p.dq) + grow * p.dq_rs + (int64_t)head * p.dq_hs + dcol;
                    f32x4 a = p.f32_store ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<f32x4*>(dqp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] += dq[mt][4 * g4 + e] * p.scale;
                    *reinterpret_cast<f32x4*>(dqp) = a;
                } else {
                    __bf16* dqp = reinterpret_cast<__bf16*>(p.dq) + grow * p.dq_rs + (int64_t)head * p.dq_hs + dcol;
                    bf16x4 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (__bf16)(dq[mt][4 * g4 + e] * p.scale);
                    *reinterpret_cast<bf16x4*>(dqp) = a;
                }
            }
    }
}

// dQ kernel choice: the ping-pong kernel (attn_bwd_dq_pp.hip) unless the environment variable
// LVATTN_DQ_KERNEL=lockstep asks for the one in this file.
static int g_dq_kernel = -1, g_dkdv_kernel = -1;   // -1: environment not read yet; 0 default kernel, 1 the one in this file
static bool dq_pingpong() {
    if (g_dq_kernel < 0) {
        const char* e = getenv("LVATTN_DQ_KERNEL");
        g_dq_kernel = (e && !strcmp(e, "lockstep")) ? 1 : 0;
    }
    return g_dq_kernel == 0;
}

// dK/dV kernel choice.  Left to the library: the gap-scheduled one-wave kernel (attn_bwd_dkdv_g.hip) at head dim 128
// (same-box A/B, 131,072 tokens, forward + backward: one segment 642.5 -> 619.5 ms, packed segments 73.4 -> 71.1 ms, packed
// at 5 / 1 heads 10.00 -> 9.62 ms), the role-split kernel (attn_bwd_dkdv_rs.hip) at head dim 64 and for descriptors with a
// band.  The environment variable LVATTN_DKDV_KERNEL = rs / g / single forces one.
static int dkdv_choice() {   // 0 the library's choice, 1 the first-generation kernel in this file, 2 gap-scheduled (d = 128), 3 role-split
    if (g_dkdv_kernel < 0) {
        const char* e = getenv("LVATTN_DKDV_KERNEL");
        g_dkdv_kernel = (e && !strcmp(e, "single")) ? 1 : (e && !strcmp(e, "g")) ? 2 : (e && !strcmp(e, "rs")) ? 3 : 0;
    }
    return g_dkdv_kernel;
}
// run-time override of the two choices above: 0 default, 1 the kernel in this file (dkdv: 2 / 3 as in dkdv_choice), < 0 leave as it is
void lv_select_bwd_kernels(int dq, int dkdv) {
    if (dq >= 0) g_dq_kernel = dq;
    if (dkdv >= 0) g_dkdv_kernel = dkdv;
}

template <int D>
static void launch_d(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    dim3 block(512);
    const bool banded = (causal == 2);     // descriptors with a band: the role-split dK/dV and the ping-pong dQ kernels read it
    causal = causal != 0;
    if (p.n_blocks > 0 && banded) {
        lv_launch_bwd_dkdv_rs(p, D, causal, accum, stream);
    } else if (p.n_blocks > 0 && (dkdv_choice() == 2 || dkdv_choice() == 0) && D == 128 && (int64_t)p.hq * p.lse_stride * 4 <= 0xffffffffLL) {
        // (that kernel addresses the [hq, lse_stride] row constants with 32-bit byte offsets)
        lv_launch_bwd_dkdv_g(p, causal, accum, stream);
    } else if (p.n_blocks > 0 && dkdv_choice() != 1) {
        lv_launch_bwd_dkdv_rs(p, D, causal, accum, stream);
    } else if (p.n_blocks > 0) {
        dim3 grid((unsigned)((int64_t)p.n_blocks * p.hkv));
        dim3 block(256);
        size_t lds = (size_t)LV_BWD_RING * (2 * 32 * D * 2 + 4 * 256);  // ring slots of (Q slice, dO slice, row constants)
        if (causal) {
            if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, true, false>), grid, block, lds, stream, p);
        } else {
            if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, false, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, false, false>), grid, block, lds, stream, p);
        }
    }
    if (p.n_qblocks > 0 && (banded || dq_pingpong())) {
        lv_launch_bwd_dq_pp(p, D, causal, accum, stream);
    } else if (p.n_qblocks > 0) {
        dim3 grid((unsigned)((int64_t)p.n_qblocks * p.hq));
        size_t lds = (size_t)4 * 64 * D * 2;
        if (causal) {
            if (accum) hipLaunchKernelGGL((lv_bwd_dq_kernel<D, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dq_kernel<D, true, false>), grid, block, lds, stream, p);
        } else {
            if (accum) hipLaunchKernelGGL((lv_bwd_dq_kernel<D, false, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dq_kernel<D, false, false>), grid, block, lds, stream, p);
        }
    }
}

int lv_launch_bwd(const BwdParams& p, int d, int causal, int accum, hipStream_t stream) {
    if (d == 128) launch_d<128>(p, causal, accum, stream);
    else launch_d<64>(p, causal, accum, stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
