// Flash-style attention forward for gfx950, "ping-pong" form on the 16x16x32 bf16 MFMA shape (d = 128).
//
// Same structure as attn_fwd_pp.hip -- one workgroup = 8 waves = one 256-row query block of one query
// head, wave w owns rows 32w .. 32w+31, waves 0-3 (group A) and 4-7 (group B) run the same two-phase
// loop one barrier apart, K / V tiles of 64 keys arrive by direct global -> LDS loads into a 3-slot ring
// -- but every product is issued as v_mfma_f32_16x16x32_bf16.  Cycles per FLOP are the same as for the
// 32x32x16 shape; the chip holds a higher clock on this one under the same load, which is where the
// kernel's time goes (it is limited by the power the matrix pipes draw, not by issue slots).
//
// Layouts (lane = 16 g + n, g = 0..3, n = 0..15), per 64-key tile and wave:
//   S^T[kb][qb] = K[16 keys kb] Q[16 rows qb]^T : A = K rows from LDS (lane: key n, d 32ks+8g..+8),
//                 B = Q fragments in registers (lane: row 16qb+n, d 32ks+8g..+8);
//                 accumulator f32x4: lane holds row 16qb+n, keys 16kb + 4g + i.
//   O^T[db][qb] += V[32 keys j][16 cols db]^T P^T : B = P straight from the S^T accumulators of key blocks 2j and
//                 2j+1 (k-slot 8g+e <-> key 32j + 16(e/4) + 4g + e%4), A = V read column-wise with two
//                 ds_read_b64_tr_b16 in the same key order; accumulator: row 16qb+n, cols 16db + 4g + i.
// A query row therefore lives on the four lanes n, n+16, n+32, n+48: row sums are kept as per-lane partial
// sums and added up once at the end; the row maximum needs two cross-lane steps, and only the exact
// softmax pass (first tile, masked tiles, outliers) computes it.
#include "internal.h"

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_PP16_PRIO
#define LV_PP16_PRIO 2   // s_setprio level of a wave in its matrix phase (0 in the vector phase)
#endif

// Swizzle of a [rows][128 x bf16] LDS tile for this kernel's two kinds of read: chunk c of row r lives at
// 256 r + 16 (c ^ f(r)), f(r) = ((r & 7) << 1) | ((r >> 3) & 1).  The 16-row x one-chunk reads of the A
// operand (ds_read_b128, lane = row) and the 8-row x 32-byte transposed reads (ds_read_b64_tr_b16, 4 rows per
// 16-lane group) both touch every bank once per pass.
__device__ __forceinline__ int swz16(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }
__device__ __forceinline__ int lds_off16(int row, int chunk) { return row * 256 + ((chunk ^ swz16(row)) << 4); }

__device__ __forceinline__ void pp16_dma16(__amdgpu_buffer_rsrc_t rs, int voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ float quad_max(float v) {   // over the four lanes n, n+16, n+32, n+48
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

template <bool CAUSAL>
__global__ __launch_bounds__(512) void lv_fwd_pp16_kernel(FwdParams p) {
    constexpr int D = 128;
    constexpr int BN = 64;            // keys per tile
    constexpr int KS = D / 32;        // k-steps of the QK^T product (4)
    constexpr int NDB = D / 16;       // 16-wide d sub-tiles of the PV product (8)
    constexpr int ROWB = D * 2;
    constexpr int TILEB = BN * ROWB;
    constexpr int CPR = D / 8;
    constexpr int NS = 3;
    constexpr int SLOTB = 2 * TILEB;  // [K tile][V tile]
    constexpr int RPP = 64 / CPR;
    constexpr int PPW = (TILEB / 1024) / 4;
    constexpr float RESCALE_LOG2 = 8.0f;
    constexpr float OPTIMISTIC_MAX = 4096.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* const lds0 = (lds_char*)smem;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grp_b = (w >= 4);
    const int wg = w & 3;
    const int n = lane & 15;
    const int g = lane >> 4;

    // Workgroup i runs on XCD i % 8: the query heads of one (KV head, query block) unit take consecutive slots of
    // one XCD and share its L2 (see attn_fwd_pp.hip); the grid is padded to whole rounds of 8 units.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int hg = slot % p.group;
    const int unit = (slot / p.group) * 8 + xcd;
    if (unit >= p.n_blocks * p.hkv) return;
    const int blk = unit % p.n_blocks, kvh = unit / p.n_blocks;
    const int head = kvh * p.group + hg;

    const AttnDesc dsc = lv_fwd_desc(p, blk, 256, CAUSAL);
    const int q_row0 = dsc.a_row0, q_rows = dsc.a_rows;
    const int k_row0 = dsc.b_row0, k_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;

    // ---- Q fragments (B operand of S^T): lane holds Q[row 32w + 16qb + n][32ks + 8g .. +8]
    int qi[2], limit[2];
    bf16x8 qf[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        qi[qb] = 32 * w + 16 * qb + n;
        const int q_row = q_row0 + (qi[qb] < q_rows ? qi[qb] : q_rows - 1);
        const __bf16* qptr = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * g;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *reinterpret_cast<const bf16x8*>(qptr + 32 * ks);
        limit[qb] = qi[qb] + shift;  // visible keys (relative to k_row0): j <= limit
        if (limit[qb] > k_len - 1 || qi[qb] >= q_rows) limit[qb] = k_len - 1;
    }
    int wave_kmax = 32 * w + 32 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    const int my_nkt = (32 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging: group A brings K tiles, group B V tiles (as in attn_fwd_pp.hip; the swizzle goes on the source)
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
        const int ch16 = lds_off16(row, lane % CPR) - row * ROWB;
        voff_full[j] = row * (int)(x_rs * 2) + ch16;
        voff_part[j] = min(row, rows_last - 1) * (int)(x_rs * 2) + ch16;
    }
    int ld_t = 0;
    uint32_t ld_dst = lds_base + (grp_b ? SLOTB + TILEB : 0) + PPW * wg * 1024;
    const uint32_t ld_wrap = lds_base + NS * SLOTB;
    auto issue_tile = [&]() __attribute__((always_inline)) {
        int rows = rows_avail - ld_t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        const uint32_t rec = rows ? (uint32_t)((rows - 1) * x_rs * 2 + ROWB) : 0u;
        auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xbase + ld_t * x_tile_bytes), 0, rec, 0x00020000);
        const bool partial = rows < BN;
#pragma unroll
        for (int j = 0; j < PPW; ++j) pp16_dma16(rs, partial ? voff_part[j] : voff_full[j], ld_dst + j * 1024);
        ++ld_t;
        ld_dst = (ld_dst + SLOTB >= ld_wrap) ? ld_dst + SLOTB - NS * SLOTB : ld_dst + SLOTB;
    };

    // ---- LDS read addresses in slot 0
    lds_char* k_ptr[KS];      // K rows: row 16kb + n, chunk 4ks + g        (+ 16 kb ROWB)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = lds0 + lds_off16(n, 4 * ks + g);
    const int q4 = n >> 2, p4 = n & 3;
    lds_char* v_ptr[NDB];     // V columns 16db .. +16 of rows 4g + q4      (+ (32j + 16hi) ROWB)
#pragma unroll
    for (int db = 0; db < NDB; ++db) v_ptr[db] = lds0 + TILEB + lds_off16(4 * g + q4, 2 * db + (p4 >> 1)) + 8 * (p4 & 1);

    f32x4 o[NDB][2];
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
    const float c = p.scale_log2e;
    f32x4 s[4][2];       // [key block][row block]
    bf16x8 pb[2][2];     // P^T operands: [32-key group][row block]

#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qb][ks]));

    if (nkt > 0) {
        issue_tile();
        issue_tile();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // S_0 (both groups at once)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            s[kb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            s[kb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *(LDS_PTR(bf16x8, k_ptr[ks] + kb * 16 * ROWB));
                s[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[kb][0], 0, 0, 0);
                s[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[kb][1], 0, 0, 0);
            }
        }
        int sb = 0;  // byte offset of the slot of the NEXT matrix phase
        if (grp_b) asm volatile("s_barrier" ::: "memory");
        for (int t = 0; t < nkt; ++t) {
            // ================= VALU(t): softmax of S_t -> P_t =================
            if (t < my_nkt) {
                const int kb0 = t * BN;
                const bool need_mask = (kb0 + BN - 1 > 32 * w + shift) || (kb0 + BN > k_len);
                // optimistic pass against the running maximum, exact pass only when needed (see attn_fwd_pp.hip)
                bool exact = need_mask || __any(fminf(m_run[0], m_run[1]) < -1e29f);
                if (!exact) {
                    float rs[2];
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        const float mc = m_run[qb] * c;
                        rs[qb] = 0.f;
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float pv = fast_exp2(__builtin_fmaf(s[kb][qb][i], c, -mc));
                                rs[qb] += pv;
                                pb[kb >> 1][qb][4 * (kb & 1) + i] = (__bf16)pv;
                            }
                    }
                    if (__any(!(fmaxf(rs[0], rs[1]) <= OPTIMISTIC_MAX))) exact = true;   // (also catches inf / nan)
                    else { l_run[0] += rs[0]; l_run[1] += rs[1]; }
                }
                if (exact) {
                    if (need_mask) {
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int j = kb0 + 16 * kb + 4 * g + i;
                                    s[kb][qb][i] = (j > limit[qb]) ? -INFINITY : s[kb][qb][i];
                                }
                    }
                    float m_new[2];
                    bool grow = false;
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        float mx = s[0][qb][0];
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[kb][qb][i]);
                        mx = quad_max(mx);
                        m_new[qb] = fmaxf(m_run[qb], mx);
                        grow = grow || ((m_new[qb] - m_run[qb]) * c > RESCALE_LOG2);
                    }
                    // deferred rescale: the running maxima are raised only when some row of the wave grew by > 2^8
                    if (__any(grow)) {
#pragma unroll
                        for (int qb = 0; qb < 2; ++qb) {
                            const float alpha = fast_exp2((m_run[qb] - m_new[qb]) * c);
                            m_run[qb] = m_new[qb];
                            l_run[qb] *= alpha;
#pragma unroll
                            for (int db = 0; db < NDB; ++db)
#pragma unroll
                                for (int i = 0; i < 4; ++i) o[db][qb][i] *= alpha;
                        }
                    }
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb) {
                        const float mc = m_run[qb] * c;
                        float rs = 0.f;
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float pv = fast_exp2(__builtin_fmaf(s[kb][qb][i], c, -mc));
                                rs += pv;
                                pb[kb >> 1][qb][4 * (kb & 1) + i] = (__bf16)pv;
                            }
                        l_run[qb] += rs;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                        for (int i = 0; i < 8; ++i) pb[j][qb][i] = (__bf16)0.f;
            }
            issue_tile();  // K (group A) or V (group B) tile t + 2
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
            __builtin_amdgcn_sched_barrier(0);

            // ================= MFMA(t+1): S_{t+1} = K_{t+1} Q^T, O += V_t P_t =================
            __builtin_amdgcn_s_setprio(LV_PP16_PRIO);
            sb = (sb + SLOTB == NS * SLOTB) ? 0 : sb + SLOTB;
            // Four groups of 16 MFMAs.  Group G: O^T[db][*] += V[keys of 32-key group G/2][cols db]^T P^T for the four
            // db of its half, and all four k-steps of S^T[key block G][*].  The fragments of group G+1 are read from
            // LDS while group G runs (two register sets).
            bf16x8 vfr[2][4], kfr[2][KS];
            auto load_group = [&](int G, int set) __attribute__((always_inline)) {
                const int j = G >> 1, dh = G & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int db = 4 * dh + i;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[db] + sb + (32 * j) * ROWB));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[db] + sb + (32 * j + 16) * ROWB));
                    bf16x8& vf = vfr[set][i];
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kfr[set][ks] = *(LDS_PTR(bf16x8, k_ptr[ks] + sb + G * 16 * ROWB));
            };
            load_group(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                const int j = G >> 1, dh = G & 1, set = G & 1;
                if (G < 3) load_group(G + 1, set ^ 1);
                s[G][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                s[G][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // alternate the accumulation targets so that consecutive MFMAs are independent
                    o[4 * dh + i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[set][i], pb[j][0], o[4 * dh + i][0], 0, 0, 0);
                    s[G][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[set][i], qf[0][i], s[G][0], 0, 0, 0);
                    o[4 * dh + i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[set][i], pb[j][1], o[4 * dh + i][1], 0, 0, 0);
                    s[G][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[set][i], qf[1][i], s[G][1], 0, 0, 0);
                }
                if (G < 3) {
                    // issue order: the next group's 12 reads spread between this group's 16 MFMAs
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
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
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float l_tot = quad_sum(l_run[qb]);
        const float inv = (l_tot > 0.f) ? 1.f / l_tot : 0.f;
        if (qi[qb] < q_rows) {
            __bf16* optr = p.o + (int64_t)(q_row0 + dsc.out_delta + qi[qb]) * p.o_rs + (int64_t)head * p.o_hs + 4 * g;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                bf16x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (__bf16)(o[db][qb][e] * inv);
                *reinterpret_cast<bf16x4*>(optr + 16 * db) = ov;
            }
            if (g == 0) {
                const float lse = (l_tot > 0.f) ? (m_run[qb] * p.scale + __logf(l_tot)) : -INFINITY;
                p.lse[(int64_t)head * p.lse_stride + q_row0 + dsc.out_delta + qi[qb]] = lse;
            }
        }
    }
}

int lv_launch_fwd_pp16(const FwdParams& p, int causal, hipStream_t stream) {
    dim3 grid((unsigned)((((int64_t)p.n_blocks * p.hkv + 7) / 8) * 8 * p.group)), block(512);
    const size_t lds = (size_t)3 * 2 * 64 * 128 * 2;  // 3 slots x (K tile + V tile)
    if (causal) hipLaunchKernelGGL((lv_fwd_pp16_kernel<true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_fwd_pp16_kernel<false>), grid, block, lds, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
