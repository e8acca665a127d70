// Flash-style attention backward for gfx950 (MI355X).
//
// lv_bwd_delta_kernel : delta[h][row] = sum_d dO * O   (fp32)
// lv_bwd_kernel       : one workgroup = 4 waves = one 128-key block of one K/V head.
//   Wave w keeps dK^T and dV^T of its 32 keys in accumulators while the workgroup
//   sweeps the query heads of the group x 32-row query slices.  Per slice and wave:
//     S  = Q K^T        (A = Q rows from LDS,  B = K fragments in registers)   key on the lane
//     P  = exp2(S*c - lse*log2e)
//     dV^T += dO^T P    (A = dO read column-wise with ds_read_b64_tr_b16, B = P registers)
//     dP = dO V^T       (A = dO rows,          B = V fragments in registers)
//     dS = P * (dP - delta) * scale
//     dK^T += Q^T dS    (A = Q read column-wise, B = dS registers)
//     dS^T -> LDS (bf16), barrier, then the workgroup's 128 keys are contracted:
//     dQ[:, 32w..32w+31] = dS K  (A = dS via transposed read, B = K tile in LDS read
//                                 column-wise) and added to the fp32 dq buffer with
//                                 global atomics (two 128-byte row segments per instruction).
#include "internal.h"

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
template <int D, bool CAUSAL, bool ACCUM_F32>
__global__ __launch_bounds__(256) void lv_bwd_kernel(BwdParams p) {
    constexpr int BN = 128;            // keys per workgroup
    constexpr int BQ = 32;             // query rows per slice
    constexpr int KS = D / 16;         // k-steps over d
    constexpr int MT = D / 32;         // 32-wide d sub-tiles
    constexpr int ROWB = D * 2;        // bytes per Q/dO/K row in LDS
    constexpr int CPR = D / 8;         // 16-byte chunks per row
    constexpr int QTILE = BQ * ROWB;   // bytes of a Q (or dO) slice
    constexpr int NLDQ = (BQ * CPR) / 256;   // 16-byte loads per thread per slice (2 or 1)
    constexpr int NLDK = (BN * CPR) / 256;   // per thread for the K tile (8 or 4)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [K tile BN x D][Q buf0][Q buf1][dO buf0][dO buf1][dS^T BN x 32 bf16][lse2 2x32 f32][delta 2x32 f32]
    char* const ldsK = smem;
    char* const ldsQ = ldsK + BN * ROWB;
    char* const ldsO = ldsQ + 2 * QTILE;
    char* const ldsS = ldsO + 2 * QTILE;
    float* const ldsL = reinterpret_cast<float*>(ldsS + BN * 64);
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

    // ---- K tile -> LDS (for dQ), K / V fragments -> registers (B operands, key = 32w + r)
    {
        const __bf16* kb = p.k + (int64_t)kvh * p.k_hs;
#pragma unroll
        for (int i = 0; i < NLDK; ++i) {
            int c = tid + 256 * i;
            int row = c / CPR, ch = c % CPR;
            int grow = k_row0 + (row < k_rows ? row : k_rows - 1);
            u32x4 v = *reinterpret_cast<const u32x4*>(kb + (int64_t)grow * p.k_rs + ch * 8);
            *reinterpret_cast<u32x4*>(ldsK + lds_offd<D>(row, ch)) = v;
        }
    }
    const int kj = 32 * w + r;  // this lane's key inside the block
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

    // ---- staging of a (head, slice): Q, dO rows and the per-row lse*log2e / delta
    int sq_row[NLDQ], sq_col[NLDQ], sq_lds[NLDQ];
#pragma unroll
    for (int i = 0; i < NLDQ; ++i) {
        int c = tid + 256 * i;
        sq_row[i] = c / CPR;
        int ch = c % CPR;
        sq_col[i] = ch * 8;
        sq_lds[i] = lds_offd<D>(sq_row[i], ch);
    }
    u32x4 qreg[NLDQ], oreg[NLDQ];
    float lreg = 0.f, dreg = 0.f;
    auto issue_loads = [&](int it) {
        int hg = it / nslice, sl = it - hg * nslice;
        int head = kvh * p.group + hg;
#pragma unroll
        for (int i = 0; i < NLDQ; ++i) {
            int ql = sl * BQ + sq_row[i];
            int grow = q_row0 + (ql < q_len ? ql : q_len - 1);
            qreg[i] = *reinterpret_cast<const u32x4*>(p.q + (int64_t)grow * p.q_rs + (int64_t)head * p.q_hs + sq_col[i]);
            oreg[i] = *reinterpret_cast<const u32x4*>(p.dout + (int64_t)grow * p.do_rs + (int64_t)head * p.do_hs + sq_col[i]);
        }
        if (tid < BQ) {
            int ql = sl * BQ + tid;
            int grow = q_row0 + (ql < q_len ? ql : q_len - 1);
            lreg = p.lse[(int64_t)head * p.lse_stride + grow] * 1.4426950408889634f;
            dreg = p.delta[(int64_t)head * p.lse_stride + grow];
        }
    };
    auto write_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLDQ; ++i) {
            *reinterpret_cast<u32x4*>(ldsQ + buf * QTILE + sq_lds[i]) = qreg[i];
            *reinterpret_cast<u32x4*>(ldsO + buf * QTILE + sq_lds[i]) = oreg[i];
        }
        if (tid < BQ) {
            ldsL[buf * BQ + tid] = lreg;
            ldsD[buf * BQ + tid] = dreg;
        }
    };

    // ---- LDS read offsets
    int a_off[KS];  // row reads of Q / dO: row r, chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_off[ks] = lds_offd<D>(r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    int t_off[MT][2];  // transposed reads of Q / dO (rows = queries in accumulator order)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh)
            t_off[mt][jh] = lds_offd<D>(8 * jh + 4 * h + q4, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);
    // dQ: A = dS[q][key] from dS^T [key][32 q] (64-byte rows): keys 16ks + 8h + 4jh + q4, queries 16gl + i16
    int s_off[2];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) s_off[jh] = (8 * h + 4 * jh + q4) * 64 + 32 * gl + 8 * p4;
    //     B = K[key][d = 32w' + r] read column-wise from the K tile; this wave covers d columns
    //     [dq_c0, dq_c0 + 32): with D = 128 wave w takes 32w; with D = 64 waves 0,1 take 32w and
    //     waves 2,3 split the keys instead (see below).
    constexpr int DQ_KSPLIT = (D == 128) ? 1 : 2;       // ways the 128 keys are split between waves
    const int dq_mt = (D == 128) ? w : (w & 1);          // 32-wide d sub-tile of this wave
    const int dq_kpart = (D == 128) ? 0 : (w >> 1);      // which part of the keys
    int kq_off[2];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh)
        kq_off[jh] = lds_offd<D>(8 * h + 4 * jh + q4, 4 * dq_mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1);

    f32x16 dk[MT], dv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dk[mt][i] = 0.f; dv[mt][i] = 0.f; }
    const float c = p.scale_log2e;

    if (niter > 0) {
        issue_loads(0);
        write_lds(0);
    }
    __syncthreads();

    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1;
        const bool more = (it + 1 < niter);
        const int hg = it / nslice, sl = it - hg * nslice;
        const int head = kvh * p.group + hg;
        const int i0 = sl * BQ;  // first query row of the slice, relative to q_row0
        if (more) issue_loads(it + 1);

        const char* qb = ldsQ + cur * QTILE;
        const char* ob = ldsO + cur * QTILE;
        // does any (query, key) pair of this wave survive the mask?  keys 32w..32w+31
        const bool active = (32 * w <= i0 + BQ - 1 + shift) && (32 * w < k_rows);
        bf16x8 dsf[2];
        if (active) {
            // ---- S = Q K^T
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 qa = *reinterpret_cast<const bf16x8*>(qb + a_off[ks]);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[ks], s, 0, 0, 0);
            }
            // ---- P = exp2(S*c - lse2), masked
            f32x4 l4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) l4[g] = *reinterpret_cast<const f32x4*>(ldsL + cur * BQ + 8 * g + 4 * h);
            const bool need_mask = (32 * w + 31 > i0 + shift) || (i0 + BQ > q_len) || (32 * w + 32 > k_rows);
            f32x16 pr;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float pv = fast_exp2(__builtin_fmaf(s[i], c, -l4[i >> 2][i & 3]));
                pr[i] = pv;
            }
            if (need_mask) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int ii = i0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    bool dead = (kj > ii + shift) || (ii >= q_len) || (kj >= k_rows);
                    pr[i] = dead ? 0.f : pr[i];
                }
            }
            bf16x8 pf[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) pf[i >> 3][i & 7] = (__bf16)pr[i];
            // ---- dV^T += dO^T P
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* base = ob + 16 * s2 * ROWB;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + t_off[mt][0]));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + t_off[mt][1]));
                    bf16x8 af;
                    af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                    af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
                    dv[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf[s2], dv[mt], 0, 0, 0);
                }
            // ---- dP = dO V^T
            f32x16 dp;
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 oa = *reinterpret_cast<const bf16x8*>(ob + a_off[ks]);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, vf[ks], dp, 0, 0, 0);
            }
            // ---- dS = P * (dP - delta) * scale
            f32x4 d4[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) d4[g] = *reinterpret_cast<const f32x4*>(ldsD + cur * BQ + 8 * g + 4 * h);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float ds = pr[i] * (dp[i] - d4[i >> 2][i & 3]) * p.scale;
                dsf[i >> 3][i & 7] = (__bf16)ds;
            }
            // ---- dK^T += Q^T dS
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const char* base = qb + 16 * s2 * ROWB;
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + t_off[mt][0]));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + t_off[mt][1]));
                    bf16x8 af;
                    af[0] = lo[0]; af[1] = lo[1]; af[2] = lo[2]; af[3] = lo[3];
                    af[4] = hi[0]; af[5] = hi[1]; af[6] = hi[2]; af[7] = hi[3];
                    dk[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, dsf[s2], dk[mt], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) { dsf[0][i] = (__bf16)0.f; dsf[1][i] = (__bf16)0.f; }
        }
        // ---- dS^T -> LDS: row = key kj, 4 consecutive queries (8g + 4h ..) per 8-byte store
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bf16x4 v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) v4[e] = dsf[g >> 1][4 * (g & 1) + e];
            *reinterpret_cast<bf16x4*>(ldsS + kj * 64 + (8 * g + 4 * h) * 2) = v4;
        }
        __syncthreads();

        // ---- dQ slice = dS K over the block's keys; skip key ranges no wave touched
        {
            // keys beyond kmax are masked for every row of the slice
            int kmax = i0 + BQ + shift;             // exclusive bound on visible keys
            kmax = kmax < 0 ? 0 : (kmax > BN ? BN : kmax);
            if (kmax > k_rows) kmax = k_rows;
            const int nks_all = (kmax + 15) / 16;   // 16-key k-steps with any live key
            constexpr int KSTEPS = BN / 16 / DQ_KSPLIT;
            f32x16 dq;
#pragma unroll
            for (int i = 0; i < 16; ++i) dq[i] = 0.f;
            for (int ks = 0; ks < KSTEPS; ++ks) {
                int kstep = dq_kpart * KSTEPS + ks;
                if (kstep >= nks_all) break;
                const char* sb = ldsS + kstep * 16 * 64;
                const char* kb2 = ldsK + kstep * 16 * ROWB;
                bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + s_off[0]));
                bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + s_off[1]));
                bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kb2 + kq_off[0]));
                bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, kb2 + kq_off[1]));
                bf16x8 af, bf;
                af[0] = alo[0]; af[1] = alo[1]; af[2] = alo[2]; af[3] = alo[3];
                af[4] = ahi[0]; af[5] = ahi[1]; af[6] = ahi[2]; af[7] = ahi[3];
                bf[0] = blo[0]; bf[1] = blo[1]; bf[2] = blo[2]; bf[3] = blo[3];
                bf[4] = bhi[0]; bf[5] = bhi[1]; bf[6] = bhi[2]; bf[7] = bhi[3];
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, dq, 0, 0, 0);
            }
            if (dq_kpart * KSTEPS < nks_all) {
                float* dqp = p.dq + ((int64_t)(q_row0 + i0) * p.hq + head) * D + 32 * dq_mt + r;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    int ii = (i & 3) + 8 * (i >> 2) + 4 * h;
                    if (i0 + ii < q_len) atomicAdd(dqp + (int64_t)ii * p.hq * D, dq[i]);
                }
            }
        }
        if (more) write_lds(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: dK^T / dV^T accumulators (key on the lane) -> rows of dk / dv
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
                    for (int e = 0; e < 4; ++e) { a[e] += dk[mt][4 * g4 + e]; b[e] += dv[mt][4 * g4 + e]; }
                    *reinterpret_cast<f32x4*>(dkp) = a;
                    *reinterpret_cast<f32x4*>(dvp) = b;
                } else {
                    __bf16* dkp = reinterpret_cast<__bf16*>(p.dk) + grow * p.dk_rs + (int64_t)kvh * p.dk_hs + dcol;
                    __bf16* dvp = reinterpret_cast<__bf16*>(p.dv) + grow * p.dv_rs + (int64_t)kvh * p.dv_hs + dcol;
                    bf16x4 a, b;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = (__bf16)dk[mt][4 * g4 + e]; b[e] = (__bf16)dv[mt][4 * g4 + e]; }
                    *reinterpret_cast<bf16x4*>(dkp) = a;
                    *reinterpret_cast<bf16x4*>(dvp) = b;
                }
            }
    }
}

template <int D>
static size_t bwd_lds_bytes() {
    return (size_t)128 * D * 2 + 4 * 32 * D * 2 + 128 * 64 + 4 * 32 * sizeof(float);
}

template <int D>
static void launch_d(const BwdParams& p, int causal, int accum, hipStream_t stream) {
    dim3 grid((unsigned)((int64_t)p.n_blocks * p.hkv)), block(256);
    size_t lds = bwd_lds_bytes<D>();
    if (causal) {
        if (accum) hipLaunchKernelGGL((lv_bwd_kernel<D, true, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_kernel<D, true, false>), grid, block, lds, stream, p);
    } else {
        if (accum) hipLaunchKernelGGL((lv_bwd_kernel<D, false, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((lv_bwd_kernel<D, false, false>), grid, block, lds, stream, p);
    }
}

int lv_launch_bwd(const BwdParams& p, int d, int causal, int accum, hipStream_t stream) {
    if (d == 128) launch_d<128>(p, causal, accum, stream);
    else launch_d<64>(p, causal, accum, stream);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
