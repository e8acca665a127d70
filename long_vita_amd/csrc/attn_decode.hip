// Incremental (single new token) attention over a KV cache for gfx950: HBM-bound, every K / V byte
// is read exactly once.
//
// One workgroup = 4 waves = one contiguous range of keys ("split") of one (batch, KV head); its 32
// MFMA query columns are the query heads of that KV head's group (GQA: group <= 32 rows, the rest
// idle).  A 128-key K / V tile is fetched global -> registers -> LDS by all four waves; wave w then
// owns keys 32w .. 32w+31 of the tile:
//   S^T = K Q^T (8 MFMA for d = 128), softmax with the query row on the lane, O^T += V^T P^T (8 MFMA)
// exactly as in the prefill kernel, so nothing but the tile loads sits on the critical path.  The LDS
// tile is single-buffered (64 KB, two barriers per tile, the next tile waits in registers) so that
// two workgroups share a CU and 128 KB per CU are in flight.  At the end the four waves' (max, sum,
// O) are combined through LDS and the split's normalised O (fp32) and log-sum-exp go to a workspace;
// lv_decode_combine_kernel merges the splits.
#include "internal.h"

template <int D>
__device__ __forceinline__ int dlds_off(int row, int chunk) {
    if constexpr (D == 128) return lds_off128(row, chunk);
    else return lds_off64(row, chunk);
}

template <int D>
__global__ __launch_bounds__(256, 2) void lv_decode_kernel(DecodeParams p) {
    constexpr int BN = 128;
    constexpr int KS = D / 16;
    constexpr int MT = D / 32;
    constexpr int ROWB = D * 2;
    constexpr int TILEB = BN * ROWB;
    constexpr int CPR = D / 8;
    constexpr int NT = 256;
    constexpr int NLD = (BN * CPR) / NT;
    constexpr float RESCALE_LOG2 = 8.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_char;
    lds_char* const ldsK = (lds_char*)smem;
    lds_char* const ldsV = ldsK + TILEB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31;
    const int h = lane >> 5;

    const int split = blockIdx.x % p.n_split;
    const int bh = blockIdx.x / p.n_split;
    const int b = bh / p.hkv, kvh = bh % p.hkv;
    const int k0 = split * p.chunk;
    int k_len = p.k_len - k0;
    k_len = k_len > p.chunk ? p.chunk : (k_len < 0 ? 0 : k_len);

    // Q fragments: row r = query head kvh * group + r (clamped; rows >= group are never stored)
    const int qr = r < p.group ? r : p.group - 1;
    const __bf16* qptr = p.q + (int64_t)b * p.q_bs + (int64_t)(kvh * p.group + qr) * p.q_hs + 8 * h;
    bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qptr + 16 * ks);

    const char* kbase = reinterpret_cast<const char*>(p.k + (int64_t)b * p.k_bs + (int64_t)kvh * p.k_hs + (int64_t)k0 * p.k_rs);
    const char* vbase = reinterpret_cast<const char*>(p.v + (int64_t)b * p.v_bs + (int64_t)kvh * p.v_hs + (int64_t)k0 * p.v_rs);
    const int64_t k_tile_bytes = (int64_t)BN * p.k_rs * 2, v_tile_bytes = (int64_t)BN * p.v_rs * 2;
    int st_lds[NLD], st_koff[NLD], st_voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int cidx = tid + NT * i;
        const int row = cidx / CPR, ch = cidx % CPR;
        st_lds[i] = dlds_off<D>(row, ch);
        st_koff[i] = (int)(row * p.k_rs * 2) + ch * 16;
        st_voff[i] = (int)(row * p.v_rs * 2) + ch * 16;
    }
    u32x4 kreg[NLD], vreg[NLD];
    auto issue_loads = [&](int t) __attribute__((always_inline)) {
        int rows = k_len - t * BN;
        rows = rows < 0 ? 0 : (rows > BN ? BN : rows);
        // rows past the end of the split read as zero (buffer bounds check) and are masked below
        const uint32_t krec = rows ? (uint32_t)((rows - 1) * p.k_rs * 2 + ROWB) : 0u;
        const uint32_t vrec = rows ? (uint32_t)((rows - 1) * p.v_rs * 2 + ROWB) : 0u;
        auto krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase + t * k_tile_bytes), 0, krec, 0x00020000);
        auto vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase + t * v_tile_bytes), 0, vrec, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krs, st_koff[i], 0, 0));
            vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, st_voff[i], 0, 0));
        }
    };
    auto write_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            *(LDS_PTR(u32x4, ldsK + st_lds[i])) = kreg[i];
            *(LDS_PTR(u32x4, ldsV + st_lds[i])) = vreg[i];
        }
    };

    // LDS read addresses of this wave's 32 keys of the tile (same fragment layouts as the prefill kernel)
    lds_char* k_ptr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_ptr[ks] = ldsK + dlds_off<D>(32 * w + r, 2 * ks + h);
    const int i16 = lane & 15, q4 = i16 >> 2, p4 = i16 & 3, gl = (lane >> 4) & 1;
    lds_char* v_ptr[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) {
            const int row = 8 * jh + 4 * h + q4;
            v_ptr[mt][jh] = ldsV + dlds_off<D>(row, 4 * mt + 2 * gl + (p4 >> 1)) + 8 * (p4 & 1) + 32 * w * ROWB;
        }

    f32x16 oacc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[mt][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const float c = p.scale_log2e;
    const int nkt = (k_len + BN - 1) / BN;

    if (nkt > 0) issue_loads(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));

    for (int t = 0; t < nkt; ++t) {
        write_lds();            // waits for tile t's loads
        __syncthreads();
        if (t + 1 < nkt) issue_loads(t + 1);
        const int kb0 = t * BN + 32 * w;     // first key of this wave's sub-tile
        if (kb0 < k_len) {
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *(LDS_PTR(bf16x8, k_ptr[ks]));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
            }
            if (kb0 + 32 > k_len) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int j = kb0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    s[i] = (j >= k_len) ? -INFINITY : s[i];
                }
            }
            float mx = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[i]);
            mx = pair32_max(mx);
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
            bf16x8 pf[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = fast_exp2(__builtin_fmaf(s[i], c, -mc));
                rs += pv;
                pf[i >> 3][i & 7] = (__bf16)pv;
            }
            l_run += rs;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int voff = 16 * s2 * ROWB;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][0] + voff));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, v_ptr[mt][1] + voff));
                    bf16x8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], oacc[mt], 0, 0, 0);
                }
            }
        }
        __syncthreads();        // every wave is done with the tile before it is overwritten
    }

    // ---- combine the four waves: stats [4][32] (max, sum), then O through the (now free) tile memory
    float* const stat = reinterpret_cast<float*>(smem);            // [2][4][32] floats = 1 KB
    float* const obuf = reinterpret_cast<float*>(smem) + 256;      // [4 waves][MT][64 lanes][16] fp32 = 4*MT*4 KB
    const float l_w = pair32_sum(l_run);
    if (h == 0) {
        stat[w * 32 + r] = m_run;
        stat[128 + w * 32 + r] = l_w;
    }
    __syncthreads();
    float m_tot = -1e30f;
#pragma unroll
    for (int i = 0; i < 4; ++i) m_tot = fmaxf(m_tot, stat[i * 32 + r]);
    float l_tot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) l_tot += stat[128 + i * 32 + r] * fast_exp2((stat[i * 32 + r] - m_tot) * c);
    const float sc = fast_exp2((m_run - m_tot) * c);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float* dst = obuf + ((w * MT + mt) * 64 + lane) * 16;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<f32x4*>(dst + 4 * g4) = f32x4{oacc[mt][4 * g4] * sc, oacc[mt][4 * g4 + 1] * sc,
                                                             oacc[mt][4 * g4 + 2] * sc, oacc[mt][4 * g4 + 3] * sc};
    }
    __syncthreads();
    if (w < MT && r < p.group) {
        const int mt = w;       // wave w finishes the d columns 32w .. 32w+31
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const float* src = obuf + ((ww * MT + mt) * 64 + lane) * 16;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + 4 * g4);
                acc[4 * g4] += t4[0]; acc[4 * g4 + 1] += t4[1]; acc[4 * g4 + 2] += t4[2]; acc[4 * g4 + 3] += t4[3];
            }
        }
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
        // workspace: part_o [bh][split][group][D] fp32, part_lse [bh][split][group]
        float* po = p.part_o + (((int64_t)bh * p.n_split + split) * p.group + r) * D + 32 * mt + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<f32x4*>(po + 8 * g4) = f32x4{acc[4 * g4] * inv, acc[4 * g4 + 1] * inv, acc[4 * g4 + 2] * inv,
                                                            acc[4 * g4 + 3] * inv};
        if (mt == 0 && h == 0)
            p.part_lse[((int64_t)bh * p.n_split + split) * p.group + r] =
                l_tot > 0.f ? (m_tot * p.scale + __logf(l_tot)) : -INFINITY;
    }
}

// out[b][head][:] = sum_s exp(lse_s - max) O_s / sum_s exp(lse_s - max); one block per (b, head).  The splits'
// log-sum-exps are fetched once by the block (thread s <-> split s) and turned into weights in LDS, then thread c sums
// column c over the splits with independent loads: two memory round trips instead of a chain of 2 x n_split.
__global__ __launch_bounds__(128) void lv_decode_combine_kernel(DecodeParams p, int d) {
    __shared__ float w_s[LV_DECODE_MAX_SPLITS];
    __shared__ float red[2];
    const int tid = threadIdx.x;
    const int bhq = blockIdx.x;               // b * hq + head
    const int b = bhq / p.hq, head = bhq % p.hq;
    const int kvh = head / p.group, g = head % p.group;
    const int64_t bh = (int64_t)b * p.hkv + kvh;
    const float* lse = p.part_lse + bh * p.n_split * p.group + g;
    float mx = -INFINITY;
    for (int s = tid; s < p.n_split; s += 128) {
        const float v = lse[(int64_t)s * p.group];
        w_s[s] = v;
        mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    for (int s = tid; s < p.n_split; s += 128) w_s[s] = mx > -INFINITY ? __expf(w_s[s] - mx) : 0.f;
    __syncthreads();
    if (tid >= d) return;
    const float* po = p.part_o + (bh * p.n_split * p.group + g) * d + tid;
    const int64_t step = (int64_t)p.group * d;
    float acc = 0.f, wsum = 0.f;
#pragma unroll 8
    for (int s = 0; s < p.n_split; ++s) {
        const float wgt = w_s[s];
        wsum += wgt;
        acc += wgt * po[s * step];
    }
    p.o[(int64_t)b * p.o_bs + (int64_t)head * p.o_hs + tid] = (__bf16)(wsum > 0.f ? acc / wsum : 0.f);
    if (p.lse && tid == 0) p.lse[(int64_t)b * p.hq + head] = mx > -INFINITY ? mx + __logf(wsum) : -INFINITY;
}

int lv_launch_decode(const DecodeParams& p, int d, hipStream_t stream) {
    dim3 grid((unsigned)((int64_t)p.n_split * p.batch * p.hkv)), block(256);
    // K tile + V tile; the epilogue reuses it for 1 KB of statistics + 4 * (d/32) * 4 KB of partial O
    size_t lds = (size_t)2 * 128 * d * 2;
    const size_t epi = 1024 + (size_t)4 * (d / 32) * 4096;
    if (epi > lds) lds = epi;
    if (d == 128) hipLaunchKernelGGL((lv_decode_kernel<128>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((lv_decode_kernel<64>), grid, block, lds, stream, p);
    if (hipGetLastError() != hipSuccess) return -2;
    hipLaunchKernelGGL(lv_decode_combine_kernel, dim3((unsigned)(p.batch * p.hq)), dim3(128), 0, stream, p, d);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
