// Flash-style attention forward for gfx950 (MI355X): bf16 MFMA, LDS-staged K/V,
// online softmax with the query row on the lane, log-sum-exp output.
//
// One workgroup = 8 waves = one 256-row query block of one query head; wave w owns
// rows 32w .. 32w+31.  Per 64-key K/V tile every wave issues 32 v_mfma_f32_32x32x16_bf16:
//   S^T = K Q^T   (A = K rows from LDS, B = Q fragments kept in registers)
//   O^T += V^T P^T (A = V read column-wise with ds_read_b64_tr_b16, B = P^T taken
//                   straight from the S^T accumulator registers, cast to bf16)
// so the query row stays on the lane (lane & 31) for S, P and O: the row maximum /
// sum are 31 in-lane ops plus one exchange with lane ^ 32, and rescaling O is a
// per-lane multiply.  K/V tiles are fetched global -> registers early and written to
// the other LDS buffer after the tile's math (one barrier per tile).
#include "internal.h"
#include <type_traits>


template <int D>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

template <int D, bool CAUSAL, int NW>
__global__ __launch_bounds__(NW * 64, 2) void lv_fwd_kernel(FwdParams p) {
    constexpr int BN = 64;            // keys per tile
    constexpr int KS = D / 16;        // k-steps of the QK^T product
    constexpr int MT = D / 32;        // 32-wide d sub-tiles of the PV product
    constexpr int ROWB = D * 2;       // bytes per K/V row
    constexpr int TILEB = BN * ROWB;  // bytes per K or V tile
    constexpr int CPR = D / 8;        // 16-byte chunks per row
    constexpr int NT = NW * 64;       // threads per workgroup
    constexpr int NLD = (BN * CPR) / NT;  // 16-byte loads per thread per tile
    static_assert(NLD * NT == BN * CPR, "the threads share a tile's 16-byte pieces evenly");
    constexpr float RESCALE_LOG2 = 8.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K buf0][K buf1][V buf0][V buf1]
    // explicit LDS (address space 3) pointers: 32-bit address arithmetic, immediates fold
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const ldsK = (lds_char*)smem;
    lds_char* const ldsV = ldsK + 2 * TILEB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    // block id -> (kv head, descriptor, head within group); heads of a group are adjacent
    // so that the workgroups in flight stream the same K/V tiles.
    int hg, blk, kvh;
    if (p.head_per_xcd) {   // all work items of a KV head on one XCD, back to back (see lv_head_per_xcd)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int per = p.n_blocks * p.group;
        kvh = (slot / per) * 8 + xcd;
        if (kvh >= p.hkv) return;
        const int rem = slot % per;
        blk = rem / p.group;
        hg = rem % p.group;
    } else {
        const int bid = blockIdx.x;
        hg = bid % p.group;
        const int rest = bid / p.group;
        blk = rest % p.n_blocks;
        kvh = rest / p.n_blocks;
    }
    const int head = kvh * p.group + hg;

    const AttnDesc dsc = lv_fwd_desc(p, blk, NW * 32, CAUSAL);
    const int q_row0 = dsc.a_row0, q_rows = dsc.a_rows;
    const int k_row0 = dsc.b_row0, k_len = dsc.b_len;
    const int shift = CAUSAL ? dsc.shift : 0x3fffffff;

    // ---- Q fragments: B operand of S^T = K Q^T : lane holds Q[row r][16ks + 8h .. +8]
    const int qi = 32 * w + r;  // row inside the block
    int q_row = q_row0 + (qi < q_rows ? qi : q_rows - 1);
    const __bf16* qptr = p.q + (int64_t)q_row * p.q_rs + (int64_t)head * p.q_hs + 8 * h;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qptr + 16 * ks);

    // visible keys (relative to k_row0) for this lane's row: j <= limit
    int limit = qi + shift;
    if (limit > k_len - 1 || qi >= q_rows) limit = k_len - 1;  // rows past q_rows are never stored
    // keys this wave needs: up to row 32w+31
    int wave_kmax = 32 * w + 32 + shift;
    wave_kmax = wave_kmax < 0 ? 0 : (wave_kmax > k_len ? k_len : wave_kmax);
    const int nkt = (k_len + BN - 1) / BN;
    // waves whose 32 rows all lie past q_rows (short last block) only help with staging
    const int my_nkt = (32 * w < q_rows) ? (wave_kmax + BN - 1) / BN : 0;

    // ---- staging (global -> regs -> LDS) through buffer loads: the descriptor base moves
    // to the tile's first key row (scalar arithmetic), the per-lane offset is fixed, and rows
    // past the end of the key range read as zero (hardware bounds check), so the loop does no
    // per-lane address arithmetic for global memory.
    const char* kbase = reinterpret_cast<const char*>(p.k + (int64_t)kvh * p.k_hs + (int64_t)k_row0 * p.k_rs);
    const char* vbase = reinterpret_cast<const char*>(p.v + (int64_t)kvh * p.v_hs + (int64_t)k_row0 * p.v_rs);
    const int64_t k_tile_bytes = (int64_t)BN * p.k_rs * 2, v_tile_bytes = (int64_t)BN * p.v_rs * 2;
    int st_lds[NLD], st_koff[NLD], st_voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int cidx = tid + NT * i;
        int row = cidx / CPR, ch = cidx % CPR;
        st_lds[i] = lds_off<D>(row, ch);
        st_koff[i] = (int)(row * p.k_rs * 2) + ch * 16;
        st_voff[i] = (int)(row * p.v_rs * 2) + ch * 16;
    }
    const int rows_avail = min(k_len, p.k_rows_total - k_row0);  // key rows that exist from k_row0 on
    // K and V of the next tile share ONE set of staging registers: K is requested at the top of a tile and written to
    // the other LDS buffer as soon as the S product has been issued, V is requested then and written at the end.
    u32x4 sreg[NLD];
    auto issue_loads = [&](int t, bool second) {
        int rows = rows_avail - t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        // num_records: whole rows of this tile (the last row only up to the end of the head's columns)
        const int64_t rs = second ? p.v_rs : p.k_rs;
        uint32_t rec = rows ? (uint32_t)((rows - 1) * rs * 2 + ROWB) : 0u;
        auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(second ? vbase + t * v_tile_bytes : kbase + t * k_tile_bytes), 0, rec, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            sreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, second ? st_voff[i] : st_koff[i], 0, 0));
    };
    auto write_lds = [&](int buf, bool second) {
        lds_char* const dst = (second ? ldsV : ldsK) + buf * TILEB;
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            *(LDS_PTR(u32x4, dst + st_lds[i])) = sreg[i];
    };

    // ---- LDS read addresses
    // K row read: row 32*st + r, chunk 2ks + h
    lds_char* k_ptr[KS];   // absolute LDS addresses (buffer 0, sub-tile 0); the rest are immediates
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = ldsK + lds_off<D>(r, 2 * ks + h);
    // V transposed read: 16-lane group g covers d columns 32mt + 16(g&1) + i, key rows
    // 16s + 8jh + 4h + q4 ; lane 4*q4 + p4 supplies row q4, columns 4*p4 .. 4*p4+3.
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* v_ptr[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            int row = 8 * jh + 4 * h + q4;
            v_ptr[mt][jh] = ldsV + lds_off<D>(row, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);
        }

    f32x16 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[mt][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale_log2e;

    if (nkt > 0) {
        issue_loads(0, false);
        write_lds(0, false);
        issue_loads(0, true);
        write_lds(0, true);
    }
    // Retire the Q loads here: otherwise the compiler waits for them inside the loop with
    // vmcnt(7..0) in front of the first MFMAs of every tile, which also drains the K/V
    // prefetch issued at the top of the iteration.
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    __syncthreads();

    // One K/V tile of the sweep; CUR (the LDS buffer) is a compile-time constant so that every
    // LDS address is a lane-constant register plus an immediate.
    auto tile = [&](int t, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const bool more = (t + 1 < nkt);
        if (more) issue_loads(t + 1, false);

        if (t < my_nkt) {
            // ---- S^T = K Q^T, two 32-key sub-tiles
            f32x16 s[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[st][i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bf16x8 kf = *(LDS_PTR(bf16x8, k_ptr[ks] + cur * TILEB + st * 32 * ROWB));
                    s[st] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[st], 0, 0, 0);
                }
            }
            if (more) { write_lds(cur ^ 1, false); issue_loads(t + 1, true); }
            // ---- mask (only tiles that cross the causal diagonal or the key-range end)
            const int kb0 = t * BN;
            const bool need_mask = (kb0 + BN - 1 > 32 * w + shift) || (kb0 + BN > k_len);
            if (need_mask) {
#pragma unroll
                for (int st = 0; st < 2; ++st)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        int j = kb0 + 32 * st + (i & 3) + 8 * (i >> 2) + 4 * h;
                        s[st][i] = (j > limit) ? -INFINITY : s[st][i];
                    }
            }
            // ---- online softmax (row = lane & 31; the other half of the row is on lane ^ 32)
            float mx = s[0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[1][i]);
            mx = pair32_max(mx);
            // Deferred rescale: the running maximum is only raised (and O, l rescaled) when some
            // row of the wave saw its maximum grow by more than 2^RESCALE_LOG2; otherwise P is
            // computed against the old maximum and is bounded by 2^RESCALE_LOG2.
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
            bf16x8 pf[2][2];
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float pv = fast_exp2(__builtin_fmaf(s[st][i], c, -mc));
                    rs += pv;
                    pf[st][i >> 3][i & 7] = (__bf16)pv;
                }
            l_run += rs;
            // ---- O^T += V^T P^T
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int voff = cur * TILEB + (32 * st + 16 * s2) * ROWB;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][0] + voff));
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][1] + voff));
                        bf16x8 vf;
                        vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                        vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                        oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[st][s2], oacc[mt], 0, 0, 0);
                    }
                }
        }
        if (more) {
            if (t >= my_nkt) { write_lds(cur ^ 1, false); issue_loads(t + 1, true); }   // a wave with no rows here only stages
            write_lds(cur ^ 1, true);
        }
        __syncthreads();
    };
    for (int t = 0; t < nkt; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nkt) tile(t + 1, std::integral_constant<int, 1>{});
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

template <int D, int NW>
static void launch_fwd_d(const FwdParams& p0, int causal, hipStream_t stream) {
    FwdParams p = p0;
    p.head_per_xcd = lv_head_per_xcd(p) ? 1 : 0;
    dim3 grid(p.head_per_xcd ? (unsigned)(((p.hkv + 7) / 8) * 8 * p.n_blocks * p.group) : (unsigned)((int64_t)p.n_blocks * p.hq)), block(NW * 64);
    size_t lds = 4 * 64 * D * 2;
    if (causal) hipLaunchKernelGGL((lv_fwd_kernel<D, true, NW>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_fwd_kernel<D, false, NW>), grid, block, lds, stream, p);
}

int lv_launch_fwd(const FwdParams& p, int d, int causal, int waves, hipStream_t stream) {
    if (d == 128) {
        if (waves == 4) launch_fwd_d<128, 4>(p, causal, stream); else launch_fwd_d<128, 8>(p, causal, stream);
    } else {
        if (waves == 4) launch_fwd_d<64, 4>(p, causal, stream); else launch_fwd_d<64, 8>(p, causal, stream);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
