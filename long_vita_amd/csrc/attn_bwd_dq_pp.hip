// dQ kernel of the attention backward for gfx950, "ping-pong" form (see attn_fwd_pp.hip for the idea):
// the two waves that share a SIMD alternate a matrix phase and a vector phase, one barrier apart.
//
// One workgroup = 8 waves = one 256-row query block of one query head; wave w owns rows 32w .. 32w+31
// (query on the lane), waves 0-3 are group A, 4-7 group B.  Per 32-key tile:
//   VALU(t)   : dS_t^T = exp2(c S_t^T - lse2) * (dP_t^T - delta)  -> bf16          (72 vector instructions)
//   MFMA(t+1) : S_{t+1}^T = K_{t+1} Q^T,  dP_{t+1}^T = V_{t+1} dO^T,  dQ^T += K_t^T dS_t^T   (24 MFMAs)
// The matrix phase is the longer one, so a SIMD's matrix pipe goes from one wave's matrix phase straight
// into its partner's.  dQ = scale * dQ^T is written once; nothing is summed across workgroups.
//
// K/V tiles live in a 5-slot LDS ring (slot t%5 = [K tile t][V tile t]; MFMA(t+1) reads K and V of slot
// t+1 by rows and K of slot t by columns).  Group A requests K tile t+4 and group B V tile t+4 late in
// MFMA(t+1) by direct global -> LDS loads; a counted s_waitcnt vmcnt before the phase-ending barrier leaves the
// newest request in flight.
#include "internal.h"
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_PP_PRIO
#define LV_PP_PRIO 2   // s_setprio level of a wave in its matrix phase (0 in the vector phase)
#endif

template <int D>
__device__ __forceinline__ int lds_off_dq(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

// Direct global -> LDS load, 16 bytes per lane, lane i -> LDS byte lds_dst + 16 i (M0 carries lds_dst).
// As asm so that the compiler does not count it; completion is waited for by hand.
__device__ __forceinline__ void dq_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

template <int D, bool CAUSAL, bool ACCUM_F32>
__global__ __launch_bounds__(512) void lv_bwd_dq_pp_kernel(BwdParams p) {
    constexpr int BN = 32;            // keys per tile
    constexpr int KS = D / 16;
    constexpr int MT = D / 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = BN * ROWB;  // bytes per K or V tile
    constexpr int CPR = D / 8;
    constexpr int NS = 5;             // ring slots
    constexpr int SLOTB = 2 * TILEB;  // [K tile][V tile]
    constexpr int RPP = 64 / CPR;     // rows per 1-KiB staging piece
    constexpr int PPW = (TILEB / 1024) / 4;  // pieces per wave per tile: 2 (d=128) or 1 (d=64)
    constexpr int NG = 4;             // MFMA groups per matrix phase
    constexpr int SPG = KS / NG;      // S (and dP) k-steps per group: 2 or 1
    constexpr int QPG = (2 * MT) / NG;  // dQ products per group: 2 or 1
    constexpr int MPG = 2 * SPG + QPG;  // MFMAs per group: 6 or 3
    constexpr int RPG = 2 * SPG + 2 * QPG;  // LDS reads per group: 8 or 4

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
    lse2 = (lse2 == -INFINITY) ? INFINITY : lse2 * 1.4426950408889634f;  // rows without keys: P = 0
    float delta = p.delta[(int64_t)head * p.lse_stride + q_row];

    int limit = qi + shift;
    if (limit > k_len - 1 || qi >= q_rows) limit = k_len - 1;
    const int band_low = CAUSAL ? lv_band_low(dsc) : -(1 << 29);   // the band's lower end, when the descriptor has one
    const int low = qi + band_low;
    int wave_kmax = 32 * w + 32 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    const int my_nkt = (32 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging: group A brings K tiles, group B V tiles; piece j of wave wg is rows
    // RPP*(PPW*wg + j) .. +RPP-1.  The swizzle of the LDS image goes on the source address.
    const int64_t x_rs = grp_b ? p.v_rs : p.k_rs;
    const char* xbase = grp_b ? reinterpret_cast<const char*>(p.v + (int64_t)kvh * p.v_hs + (int64_t)k_row0 * p.v_rs)
                              : reinterpret_cast<const char*>(p.k + (int64_t)kvh * p.k_hs + (int64_t)k_row0 * p.k_rs);
    const int64_t x_tile_bytes = (int64_t)BN * x_rs * 2;
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);
    const int rows_last = rows_avail - ((rows_avail + BN - 1) / BN - 1) * BN;
    const int prow0 = RPP * PPW * wg + lane / CPR;
    int voff_full[PPW], voff_part[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int row = prow0 + RPP * j;
        const int ch16 = lds_off_dq<D>(row, lane % CPR) - row * ROWB;
        voff_full[j] = row * (int)(x_rs * 2) + ch16;
        voff_part[j] = min(row, rows_last - 1) * (int)(x_rs * 2) + ch16;  // rows past the end repeat the last valid one
    }
    int ld_t = 0;  // next tile to request (tile t goes to slot t % NS)
    uint32_t ld_dst = lds_base + (grp_b ? TILEB : 0) + PPW * wg * 1024;
    const uint32_t ld_wrap = lds_base + NS * SLOTB;
    auto issue_tile = [&]() __attribute__((always_inline)) {
        int rows = rows_avail - ld_t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        const uint32_t rec = rows ? (uint32_t)((rows - 1) * x_rs * 2 + ROWB) : 0u;  // past the end: nothing fetched
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + ld_t * x_tile_bytes), 0, rec, 0x00020000);
        const bool partial = rows < BN;
#pragma unroll
        for (int j = 0; j < PPW; ++j) dq_dma16(rs, partial ? voff_part[j] : voff_full[j], ld_dst + j * 1024);
        ++ld_t;
        ld_dst = (ld_dst + SLOTB >= ld_wrap) ? ld_dst + SLOTB - NS * SLOTB : ld_dst + SLOTB;
    };

    // ---- LDS read addresses in slot 0
    lds_char* k_ptr[KS];  // row reads: row r, chunk 2ks + h (K; V is TILEB further)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = lds0 + lds_off_dq<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* kt_ptr[MT][2];  // transposed reads of K: rows = keys in accumulator order, columns = d
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            kt_ptr[mt][jh] = lds0 + lds_off_dq<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);

    f32x16 dq[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) dq[mt][i] = 0.f;
    const float c = p.scale_log2e;
    f32x16 s, dp;
    bf16x8 dsf[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) { dsf[0][i] = (__bf16)0.f; dsf[1][i] = (__bf16)0.f; }

    // Retire every ordinary global load here.  The compiler does not see the staging loads, so a wait
    // it placed at the first use inside the loop would be vmcnt(0) and drain them on every tile.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(qf[ks])); asm volatile("" : "+v"(dof[ks])); }
    asm volatile("" : "+v"(lse2), "+v"(delta));

    // fragments of one MFMA group: SPG K rows, SPG V rows (slot sb_new), QPG K columns (slot sb_old)
    bf16x8 kfr[2][SPG], vfr[2][SPG], tfr[2][QPG];
    auto load_group = [&](int g, int set, int sb_new, int sb_old) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < SPG; ++i) {
            kfr[set][i] = *(LDS_PTR(bf16x8, k_ptr[SPG * g + i] + sb_new));
            vfr[set][i] = *(LDS_PTR(bf16x8, k_ptr[SPG * g + i] + sb_new + TILEB));
        }
#pragma unroll
        for (int i = 0; i < QPG; ++i) {
            const int j = QPG * g + i, s2 = j / MT, mt = j % MT;
            bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kt_ptr[mt][0] + sb_old + 16 * s2 * ROWB));
            bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kt_ptr[mt][1] + sb_old + 16 * s2 * ROWB));
            bf16x8& af = tfr[set][i];
            af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
            af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
        }
    };

    if (nkt > 0) {
        // prologue: tiles 0..3 of this group's operand, then S_0 and dP_0 (both groups at once)
        issue_tile();
        issue_tile();
        issue_tile();
        issue_tile();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 kf = *(LDS_PTR(bf16x8, k_ptr[ks]));
            bf16x8 vf = *(LDS_PTR(bf16x8, k_ptr[ks] + TILEB));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dp, 0, 0, 0);
        }
        int sb = 0;  // byte offset of the slot of the last matrix phase (tile t)
        if (grp_b) asm volatile("s_barrier" ::: "memory");
        for (int t = 0; t < nkt; ++t) {
            // ================= VALU(t): dS_t =================
            if (t < my_nkt) {
                const int kb0 = t * BN;
                const bool need_mask = (kb0 + BN - 1 > 32 * w + shift) || (kb0 + BN > k_len) || (kb0 < 32 * w + 31 + band_low);
                // all the multiply-adds, then all the exponentials, then their consumers: back to back a
                // dependent pair would stall on the first one's latency sixteen times over
                float pv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[i] = __builtin_fmaf(s[i], c, -lse2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) pv[i] = fast_exp2(pv[i]);
                __builtin_amdgcn_sched_barrier(0);
                if (need_mask) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int j = kb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                        pv[i] = (j > limit || j < low) ? 0.f : pv[i];
                        asm volatile("" : "+v"(pv[i]));  // keeps this a branch
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) dsf[i >> 3][i & 7] = (__bf16)(pv[i] * (dp[i] - delta));
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) { dsf[0][i] = (__bf16)0.f; dsf[1][i] = (__bf16)0.f; }
            }
            // The vector phase also reads the first fragment group of the coming matrix phase, whose tiles
            // were complete and visible at the barrier before this phase, so that the first MFMA after
            // the barrier does not wait for LDS.
            const int sb_old = sb;
            sb = (sb + SLOTB == NS * SLOTB) ? 0 : sb + SLOTB;
            load_group(0, 0, sb, sb_old);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PPW) : "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ================= MFMA(t+1): S_{t+1}, dP_{t+1}, dQ += K_t^T dS_t =================
            __builtin_amdgcn_s_setprio(LV_PP_PRIO);
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int set = g & 1;
                if (g + 1 < NG) load_group(g + 1, set ^ 1, sb, sb_old);
#pragma unroll
                for (int i = 0; i < SPG; ++i) {
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[set][i], qf[SPG * g + i], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[set][i], dof[SPG * g + i], dp, 0, 0, 0);
                    if (i < QPG) {
                        const int j = QPG * g + i;
                        dq[j % MT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tfr[set][i], dsf[j / MT], dq[j % MT], 0, 0, 0);
                    }
                }
                static_assert(QPG <= SPG, "one dQ product per S/dP step at most");
                if (g + 1 < NG) {
                    // issue order: the next group's reads spread between this group's MFMAs (4 reads per 3 MFMAs)
                    static_assert(RPG * 3 == MPG * 4, "four reads per three MFMAs");
#pragma unroll
                    for (int i = 0; i < MPG / 3; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                }
                if (g == NG - 2) {
                    // the staging request (K tile t+4 for group A, V tile t+4 for group B) rides late in the
                    // matrix phase: all its LDS reads are out, and its scalar work fits between the MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                    issue_tile();
                    __builtin_amdgcn_sched_barrier(0);
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

template <int D>
static void launch_dq_pp(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    dim3 grid((unsigned)((int64_t)p.n_qblocks * p.hq)), block(512);
    size_t lds = (size_t)5 * 2 * 32 * D * 2;  // 5 slots x (K tile + V tile) of 32 keys
    if (causal) {
        if (accum) hipLaunchKernelGGL((lv_bwd_dq_pp_kernel<D, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dq_pp_kernel<D, true, false>), grid, block, lds, stream, p);
    } else {
        if (accum) hipLaunchKernelGGL((lv_bwd_dq_pp_kernel<D, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_dq_pp_kernel<D, false, false>), grid, block, lds, stream, p);
    }
}

int lv_launch_bwd_dq_pp(const BwdParams& p, int d, int causal, int accum, hipStream_t stream) {
    if (p.n_qblocks <= 0) return 0;
    if (d == 128) launch_dq_pp<128>(p, causal, accum, stream);
    else launch_dq_pp<64>(p, causal, accum, stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
