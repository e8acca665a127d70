// Flash-style attention backward for gfx950 (MI355X), three kernels, no atomics:
//
// lv_bwd_delta_kernel : delta[h][row] = sum_d dO * O   (fp32)
//
// lv_bwd_dkdv_kernel  : one workgroup = 4 waves = one 128-key block of one K/V head; wave w
//   keeps the K, V fragments and the dK^T, dV^T accumulators of its 32 keys in registers (key on the lane) while the workgroup
//   sweeps the query heads of the group x 32-row query slices staged in LDS:
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
#include <type_traits>

typedef __attribute__((address_space(3))) char lds_char;

#ifndef LV_BWD_KT
#define LV_BWD_KT 1   // key sub-tiles per wave of the dK/dV kernel; LV_BWD_BLOCK_N = 128 * LV_BWD_KT
#endif
static_assert(128 * LV_BWD_KT == LV_BWD_BLOCK_N_INTERNAL, "keep in step with LV_BWD_BLOCK_N in lvattn.h");

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
    constexpr int NT = NW * 64;
    constexpr int KT = LV_BWD_KT;      // 32-key sub-tiles per wave (32*KT keys per wave, 128*KT per workgroup)
    constexpr int BQ = 32;             // query rows per slice
    constexpr int KS = D / 16;         // k-steps over d
    constexpr int MT = D / 32;         // 32-wide d sub-tiles
    constexpr int ROWB = D * 2;        // bytes per Q/dO row in LDS
    constexpr int CPR = D / 8;         // 16-byte chunks per row
    constexpr int QTILE = BQ * ROWB;   // bytes of a Q (or dO) slice
    constexpr int NLD = (2 * BQ * CPR) / NT;  // 16-byte loads per thread per slice (Q and dO together)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [Q buf0][dO buf0][Q buf1][dO buf1][lse2 2x32 f32][delta 2x32 f32]
    lds_char* const lds0 = (lds_char*)smem;
    constexpr int SLICEB = 2 * QTILE;  // Q + dO of one slice
    float* const ldsL = reinterpret_cast<float*>(smem + 2 * SLICEB);
    float* const ldsD = ldsL + 2 * BQ;

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

    // ---- K / V fragments -> registers (B operands; this lane's keys = 64w + 32kt + r)
    const int kbase_w = 32 * KT * w;
    bf16x8 kf[KT][KS], vf[KT][KS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        int kj = kbase_w + 32 * kt + r;
        int grow = k_row0 + (kj < k_rows ? kj : k_rows - 1);
        const __bf16* kp = p.k + (int64_t)grow * p.k_rs + (int64_t)kvh * p.k_hs + 8 * h;
        const __bf16* vp = p.v + (int64_t)grow * p.v_rs + (int64_t)kvh * p.v_hs + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[kt][ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
            vf[kt][ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
        }
    }

    // ---- staging of a (head, slice): Q and dO rows, and the per-row lse*log2e / delta
    int sg_lds[NLD], sg_row[NLD], sg_col[NLD];
    bool sg_is_do[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int cidx = tid + NT * i;
        sg_is_do[i] = cidx >= BQ * CPR;
        int cc = sg_is_do[i] ? cidx - BQ * CPR : cidx;
        sg_row[i] = cc / CPR;
        int ch = cc % CPR;
        sg_col[i] = ch * 8;
        sg_lds[i] = (sg_is_do[i] ? QTILE : 0) + lds_offd<D>(sg_row[i], ch);
    }
    u32x4 sreg[NLD];
    float lreg = 0.f, dreg = 0.f;
    auto issue_loads = [&](int it) {
        int hg = it / nslice, sl = it - hg * nslice;
        int head = kvh * p.group + hg;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int ql = sl * BQ + sg_row[i];
            int grow = q_row0 + (ql < q_len ? ql : q_len - 1);
            const __bf16* src = sg_is_do[i] ? p.dout + (int64_t)grow * p.do_rs + (int64_t)head * p.do_hs
                                            : p.q + (int64_t)grow * p.q_rs + (int64_t)head * p.q_hs;
            sreg[i] = *reinterpret_cast<const u32x4*>(src + sg_col[i]);
        }
        if (tid < BQ) {
            int ql = sl * BQ + tid;
            int grow = q_row0 + (ql < q_len ? ql : q_len - 1);
            float l = p.lse[(int64_t)head * p.lse_stride + grow];
            lreg = (l == -INFINITY) ? INFINITY : l * 1.4426950408889634f;  // rows without keys: P = 0
            dreg = p.delta[(int64_t)head * p.lse_stride + grow];
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) *(LDS_PTR(u32x4, lds0 + buf * SLICEB + sg_lds[i])) = sreg[i];
        if (tid < BQ) {
            ldsL[buf * BQ + tid] = lreg;
            ldsD[buf * BQ + tid] = dreg;
        }
    };

    // ---- LDS read addresses (buffer 0; the other buffer and dO are immediates)
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

    f32x16 dk[KT][MT], dv[KT][MT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk[kt][mt][i] = 0.f; dv[kt][mt][i] = 0.f; }
    const float c = p.scale_log2e;

    if (niter > 0) {
        issue_loads(0);
        write_lds(0);
    }
    // retire the K/V fragment loads before the loop (no vmcnt for them inside it)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(kf[kt][ks])); asm volatile("" : "+v"(vf[kt][ks])); }
    __syncthreads();

    auto body = [&](int it, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        constexpr int QB = cur * SLICEB;          // Q of this buffer
        constexpr int OB = cur * SLICEB + QTILE;  // dO of this buffer
        const bool more = (it + 1 < niter);
        const int sl = it % nslice;
        const int i0 = sl * BQ;  // first query row of the slice, relative to q_row0
        if (more) issue_loads(it + 1);

        // does any (query, key) pair of this wave survive the mask?  keys 64w .. 64w+63
        const bool active = (kbase_w <= i0 + BQ - 1 + shift) && (kbase_w < k_rows);
        if (active) {
            const bool need_mask = (kbase_w + 32 * KT - 1 > i0 + shift) || (i0 + BQ > q_len) || (kbase_w + 32 * KT > k_rows);
            bf16x8 pf[KT][2];
            {
                // ---- S = Q K^T and P = exp2(S*c - lse2) (masked, kept as bf16), one key sub-tile
                // at a time: the Q fragments are read twice, but only 16 accumulator registers
                // are live beside the 384 that hold K, V, dK and dV.
                f32x4 l4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) l4[g] = *reinterpret_cast<const f32x4*>(ldsL + cur * BQ + 8 * g + 4 * h);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x16 sv;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sv[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 qa = *(LDS_PTR(bf16x8, a_ptr[ks] + QB));
                        sv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[kt][ks], sv, 0, 0, 0);
                    }
                    const int kj = kbase_w + 32 * kt + r;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float pv = fast_exp2(__builtin_fmaf(sv[i], c, -l4[i >> 2][i & 3]));
                        if (need_mask) {
                            int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            bool dead = (kj > ii + shift) || (ii >= q_len) || (kj >= k_rows);
                            pv = dead ? 0.f : pv;
                        }
                        pf[kt][i >> 3][i & 7] = (__bf16)pv;
                    }
                }
            }
            // ---- dV^T += dO^T P  (each transposed A fragment is read once)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][0] + OB + 16 * s2 * ROWB));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][1] + OB + 16 * s2 * ROWB));
                    bf16x8 af;
                    af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                    af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) dv[kt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf[kt][s2], dv[kt][mt], 0, 0, 0);
                }
            // ---- dP = dO V^T ; dS = P * (dP - delta) * scale, overwriting P; one sub-tile at a time
            {
                f32x4 d4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) d4[g] = *reinterpret_cast<const f32x4*>(ldsD + cur * BQ + 8 * g + 4 * h);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    f32x16 dp;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        bf16x8 oa = *(LDS_PTR(bf16x8, a_ptr[ks] + OB));
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[kt][ks], dp, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float pv = (float)pf[kt][i >> 3][i & 7];
                        pf[kt][i >> 3][i & 7] = (__bf16)(pv * (dp[i] - d4[i >> 2][i & 3]) * p.scale);
                    }
                }
            }
            // ---- dK^T += Q^T dS
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][0] + QB + 16 * s2 * ROWB));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, t_ptr[mt][1] + QB + 16 * s2 * ROWB));
                    bf16x8 af;
                    af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                    af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) dk[kt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf[kt][s2], dk[kt][mt], 0, 0, 0);
                }
        }
        if (more) write_lds(cur ^ 1);
        __syncthreads();
    };
    for (int it = 0; it < niter; it += 2) {
        body(it, std::integral_constant<int, 0>{});
        if (it + 1 < niter) body(it + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: dK^T / dV^T accumulators (key on the lane) -> rows of dk / dv
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int kj = kbase_w + 32 * kt + r;
        if (kj < k_rows) {
            const int64_t grow = k_row0 + kj;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int dcol = 32 * mt + 8 * g4 + 4 * h;
                    if constexpr (ACCUM_F32) {
                        float* dkp = reinterpret_cast<float*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                        float* dvp = reinterpret_cast<float*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                        f32x4 a = *reinterpret_cast<f32x4*>(dkp), b = *reinterpret_cast<f32x4*>(dvp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] += dk[kt][mt][4 * g4 + e]; b[e] += dv[kt][mt][4 * g4 + e]; }
                        *reinterpret_cast<f32x4*>(dkp) = a;
                        *reinterpret_cast<f32x4*>(dvp) = b;
                    } else {
                        __bf16* dkp = reinterpret_cast<__bf16*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                        __bf16* dvp = reinterpret_cast<__bf16*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                        bf16x4 a, b;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] = (__bf16)dk[kt][mt][4 * g4 + e]; b[e] = (__bf16)dv[kt][mt][4 * g4 + e]; }
                        *reinterpret_cast<bf16x4*>(dkp) = a;
                        *reinterpret_cast<bf16x4*>(dvp) = b;
                    }
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
        const int64_t grow = q_row0 + qi;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dcol = 32 * mt + 8 * g4 + 4 * h;
                if constexpr (ACCUM_F32) {
                    float* dqp = reinterpret_cast<float*>(p.dq) + grow * p.dq_rs + (int64_t)head * p.dq_hs + dcol;
                    f32x4 a = *reinterpret_cast<f32x4*>(dqp);
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
static void launch_d(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    dim3 block(512);
    if (p.n_blocks > 0) {
        dim3 grid((unsigned)((int64_t)p.n_blocks * p.hkv));
        dim3 block(256);
        size_t lds = (size_t)4 * 32 * D * 2 + 4 * 32 * sizeof(float);
        if (causal) {
            if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, true, false>), grid, block, lds, stream, p);
        } else {
            if (accum) hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, false, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((lv_bwd_dkdv_kernel<D, false, false>), grid, block, lds, stream, p);
        }
    }
    if (p.n_qblocks > 0) {
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
